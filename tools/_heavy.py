import sys, os; sys.path.insert(0,'.')
import numpy as np, torch, ctypes
import rust_tracer_amd as rta
libc = ctypes.CDLL(None)
s=rta.Scene.default(); d=s.device()
w,h=1920,1080
out = torch.zeros(64*64*4, dtype=torch.uint8, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
def timeit(reg, n=3, var=0):
    libc.setenv(b"RT_SKIP_VARIANT", str(var).encode(), 1)
    rc = d._regions([reg]); best=1e9
    for _ in range(n):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); d.render_tiles_device((w,h,1), rc, out.data_ptr(), stream, rta.RT_TRAVERSAL_SKIP); e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1)*1e3)
    return best
res=[]
for y in range(0,h,16):
    for x in range(0,w,16):
        reg=(x,min(y+16,h),min(x+16,w),y)
        res.append((timeit(reg,1),reg))
res.sort(reverse=True)
print("top 16x16 blocks (us):", [(round(t,1),r) for t,r in res[:6]])
print("median block us:", res[len(res)//2][0])
t,reg=res[0]
x0,y1,x1,y0=reg
for yy in (y0,y0+8):
    for xx in (x0,x0+8):
        r=(xx,yy+8,xx+8,yy)
        print("8x8", r, [round(timeit(r,5,v),1) for v in (0,1,2,3)])
os.environ["RT_DEBUG_STEPS"]="1"; libc.setenv(b"RT_DEBUG_STEPS", b"1", 1)
for yy in (y0,y0+8):
    for xx in (x0,x0+8):
        _,st=d.render_tiles((w,h,1),[(xx,yy+8,xx+8,yy)],rta.RT_TRAVERSAL_SKIP); print((xx,yy), st['sphere_tests'],st['bound_tests'])
print("empty-ish launch floor:", timeit((0,8,8,0),5))
