#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float v1(float x){ float y=__builtin_amdgcn_rcpf(x); float e=__builtin_fmaf(-x,y,1.0f); return __builtin_fmaf(e,y,y);}
__device__ __forceinline__ float v2(float x){ float y=v1(x); float r=__builtin_fmaf(-x,y,1.0f); return __builtin_fmaf(r,y,y);}
__device__ __forceinline__ float v0(float x){ return __builtin_amdgcn_rcpf(x);}
__global__ void k(unsigned long long *bad, unsigned *first)
{
    unsigned long long stride=(unsigned long long)gridDim.x*blockDim.x;
    unsigned long long b0=0,b1=0,b2=0;
    for (unsigned long long i=(unsigned long long)blockIdx.x*blockDim.x+threadIdx.x;i<(1ull<<32);i+=stride){
        float x=__uint_as_float((unsigned)i);
        float ax=__builtin_fabsf(x);
        if(!(ax>=0x1p-100f && ax<=0x1p100f)) continue;
        float ref=1.0f/x;
        if(__float_as_uint(v0(x))!=__float_as_uint(ref)) ++b0;
        if(__float_as_uint(v1(x))!=__float_as_uint(ref)) {++b1; atomicMin(first,(unsigned)i);}
        if(__float_as_uint(v2(x))!=__float_as_uint(ref)) {++b2; atomicMin(first+1,(unsigned)i);}
    }
    atomicAdd(bad,b0);atomicAdd(bad+1,b1);atomicAdd(bad+2,b2);
}
int main(){ unsigned long long *d; unsigned *f; hipMalloc(&d,24); hipMalloc(&f,8); hipMemset(d,0,24); hipMemset(f,0xff,8);
 k<<<8192,256>>>(d,f); hipDeviceSynchronize(); unsigned long long h[3]; unsigned hf[2]; hipMemcpy(h,d,24,hipMemcpyDeviceToHost); hipMemcpy(hf,f,8,hipMemcpyDeviceToHost);
 printf("rcp alone bad=%llu  v1(3 ops) bad=%llu first=%08x  v2(5 ops) bad=%llu first=%08x\n",h[0],h[1],hf[0],h[2],hf[1]); return 0;}
