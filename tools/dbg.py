import sys, numpy as np
sys.path.insert(0,'tests')
from hgtest import golden as G, hip, oracle as O, workloads as WL
HG=hip.load()
ctx=HG.Context(0)
for (W,H,nx,ny,A) in [(800,400,4,4,5.0),(256,64,2,2,2.0),(1030,64,2,2,2.0)]:
    img=G.lcg_image(W,H,3)
    sp,tris=WL.grid_points(W,H,nx,ny),WL.grid_triangles(nx,ny)
    dp=WL.sin_dst(sp,A,8); geom=WL.piecewise_geom(dp); ms=WL.src_min(sp)
    ctx.set_image(img); ctx.piecewise_set_mesh(sp,tris,*ms); ctx.piecewise_prepare(dp,geom)
    got=ctx.warp_inverse_piecewise()
    want,wmap,_,_=O.warp_inverse_piecewise(sp,dp,tris,img,ms[0],ms[1],*geom,taps=True)
    bad=np.any(got!=want,axis=2)
    print(W,H,geom,'bad px',bad.sum(), 'of', bad.size)
    if bad.sum():
        ys,xs=np.nonzero(bad); print(' rows',np.unique(ys)[:10],' cols', np.unique(xs)[:20], np.unique(xs//256), np.unique(xs%64)[:10])
        y,x=ys[0],xs[0]; print(' first',y,x,got[y,x],want[y,x], 'map',wmap[y*geom[2]+x])
        m=ctx.get_tri_map(fused=True); print(' map mismatches', (m!=wmap).sum())
