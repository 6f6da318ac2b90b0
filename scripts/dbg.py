import sys; sys.path.insert(0,'.')
import numpy as np, os
from ocean_model_grid_generator_amd import ocean_grid_generator as ogg
from oracle import ogg_oracle as o
Ni,Nj,lat0=5760,960,64.03160594077568
rp=np.tan(0.5*(90-lat0)*o.PI_180)
got=ogg.bipolar_cap_metrics_quad_fast(5,Ni,Nj,lat0,-300.0,rp)
for (a,b) in ((0,40),(440,480),(840,887),(887,920),(920,950),(950,960)):
    want=o.bipolar_cap_metrics_quad_fast(5,Ni,Nj,lat0,-300.0,rp,rows_per_chunk=8,j_first=a,j_last=b)
    out=[]
    for g,w,name in zip(got,want,("dx","dy","area")):
        gg,ww=g[a:b],w[a:b]
        m=ww!=0
        out.append("%s rel %.2e abs %.2e"%(name, (np.abs(gg-ww)[m]/np.abs(ww[m])).max(), np.abs(gg-ww).max()))
    print((a,b), " | ".join(out))
