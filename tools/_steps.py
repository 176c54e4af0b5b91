import sys, ctypes; sys.path.insert(0,'.')
import rust_tracer_amd as rta
libc = ctypes.CDLL(None)
s=rta.Scene.default(); d=s.device()
for var in (0,3):
    libc.setenv(b"RT_SKIP_VARIANT", str(var).encode(), 1)
    for (w,h,spp,regs) in ((1920,1080,1,None),(1920,1080,1,[(944,288,960,272)]),(1920,1080,1,[(0,8,8,0)])):
        regs = regs or [tuple(r) for r in rta.buckets(rta.RenderOptions(w,h,spp))]
        for _ in range(2): _,st=d.render_tiles((w,h,spp),regs,rta.RT_TRAVERSAL_SKIP)
        print(var, len(regs), st['device_ms'])
