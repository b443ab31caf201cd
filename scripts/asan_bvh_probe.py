import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C, sys
import rtw_amd as R
def run(items, t0=0.0, t1=0.0):
    sp = []
    for x, y, z, r, vy in items:
        s = R.Sphere.with_albedo((0.0, 0.0, 0.0), 1.0, (0.5, 0.5, 0.5))
        s.pod.center[0], s.pod.center[1], s.pod.center[2], s.pod.radius, s.pod.velocity[1] = x, y, z, r, vy
        sp.append(s)
    sc = R.Scene(sp)
    nn, depth, nbig, f16 = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = R.lib().rtw_bvh_validate(C.byref(sc.pod), float(t0), float(t1), C.byref(nn), C.byref(depth), C.byref(nbig), C.byref(f16))
    print(items[:3], '->', rc, nn.value, depth.value, nbig.value, flush=True)
inf=float('inf'); nan=float('nan')
for items in ([(0,0,0,1,0)]*3, [(inf,0,0,1,0),(0,0,0,1,0),(1,1,1,1,0)], [(nan,0,0,1,0),(0,0,0,1,0),(1,1,1,1,0)], [(0,0,0,inf,0),(0,0,0,1,0),(1,1,1,1,0)],
              [(0,0,0,nan,0),(0,0,0,1,0),(1,1,1,1,0)], [(3.4e38,0,0,3.4e38,0),(-3.4e38,0,0,1,0),(1,1,1,1,0)], [(1e30,1e30,1e30,1e30,50),(0,0,0,1e-30,0),(1,1,1,0,0)]):
    run(items); run(items, -2, 2)
