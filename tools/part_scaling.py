import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
sph = np.fromfile(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'scene_sf.bin'), dtype=rt.SPHERE_DTYPE)
rt.init(0); rt.set_world(sph)
W=3840; H=rt.height_for(W); cam=rt.camera(W,H)
buf=np.zeros((H,W,4),np.uint8)
full = rt.render(cam,W,H,1000,50,chunk_spp=rt.CHUNK_AUTO,out=buf)['stats'].kernel_ms
print("full", full)
for n in (2,4,8):
    ts=[]
    for i in range(n):
        for rep in range(2):
            st = rt.render(cam,W,H,1000,50,chunk_spp=rt.CHUNK_AUTO,tile_rows=2,part_index=i,part_count=n,out=buf)['stats']
        ts.append(st.kernel_ms)
    print("parts %d: kernel ms per part %s; max %.1f; ideal %.1f; efficiency (full/(n*max)) %.3f" % (n, ["%.1f"%t for t in ts], max(ts), full/n, full/(n*max(ts))))
