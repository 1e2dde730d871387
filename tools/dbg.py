import sys, numpy as np
sys.path.insert(0,'tests')
from hgtest import golden as G, hip, oracle as O, workloads as WL
HG=hip.load()
ctx=HG.Context(0)
for (W,H,nx,ny,A) in [(400,400,20,10,20.0),(256,64,2,2,2.0),(512,64,4,2,2.0)]:
    img=G.lcg_image(W,H,3)
    sp,tris=WL.grid_points(W,H,nx,ny),WL.grid_triangles(nx,ny)
    dp=WL.sin_dst(sp,A,8); geom=WL.piecewise_geom(dp); ms=WL.src_min(sp)
    ctx.set_image(img); ctx.piecewise_set_mesh(sp,tris,*ms); ctx.piecewise_prepare(dp,geom)
    got=ctx.warp_inverse_piecewise()
    want,wmap,_,_=O.warp_inverse_piecewise(sp,dp,tris,img,ms[0],ms[1],*geom,taps=True)
    bad=np.any(got!=want,axis=2)
    print(W,H,geom,'bad px',bad.sum(), 'of', bad.size)
    try:
        m=ctx.get_tri_map(fused=True); mm=(m!=wmap); print(' map mismatches', mm.sum())
        if mm.sum():
            idx=np.nonzero(mm)[0]; ys,xs=idx//geom[2], idx%geom[2]
            print('  rows',np.unique(ys)[:12],'cols',np.unique(xs)[:24]); i=idx[0]; print('  first',ys[0],xs[0],'got',m[i],'want',wmap[i])
    except Exception as e: print(' tap err', e)
    if bad.sum():
        ys,xs=np.nonzero(bad); print(' rows',np.unique(ys)[:10],' cols', np.unique(xs)[:20])
