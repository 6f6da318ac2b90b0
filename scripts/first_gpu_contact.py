# first GPU contact: parity of the kernels written so far against the oracle, with max-abs diffs printed
import sys, time, ctypes
sys.path.insert(0, '.')
import numpy as np
from ocean_model_grid_generator_amd import _lib as L
from oracle import ogg_oracle as orc
lib = L.load()
print("devices:", L.device_count(), L.device_name())
def rep(name, a, b):
    d = np.abs(a - b); print("%-28s shape=%-14s max|d|=%.3e  max rel=%.3e  exact=%s" % (name, a.shape, d.max(), (d/np.maximum(np.abs(b),1e-300)).max(), np.array_equal(a,b)))
# mercator
for Ni in (180, 1440, 5760):
    phi = np.array([-66.85954725*orc.PI_180, 64.05895973*orc.PI_180]); ys = np.zeros(2, dtype=np.int64)
    L.call("ogg_y_mercator_rounded", Ni, 2, L.ptr(phi), L.ptr(ys)); print("ystar", Ni, ys, orc.y_mercator_rounded(Ni, phi))
    y = np.arange(ys[0], ys[1]+1).astype(np.float64); out = np.empty_like(y)
    L.call("ogg_phi_mercator", Ni, y.size, L.ptr(y), L.ptr(out)); rep("phi_mercator %d"%Ni, out, orc.phi_mercator(Ni, np.arange(ys[0], ys[1]+1)))
# latlon + midas + angle
xo, yo = orc.generate_mercator_grid(1440, -68.0, 65.0, -300.0, 360, 2.0, True, False)
nj1, ni1 = xo.shape
dx=np.empty((nj1,ni1-1)); dy=np.empty((nj1-1,ni1)); ar=np.empty((nj1-1,ni1-1)); an=np.empty((nj1,ni1))
L.call("ogg_grid_metrics_midas", nj1, ni1, L.ptr(xo), L.ptr(yo), 6371.0e3, 1, L.ptr(dx), L.ptr(dy), L.ptr(ar))
L.call("ogg_angle_x", nj1, ni1, L.ptr(xo), L.ptr(yo), L.ptr(an))
odx, ody, oar = orc.generate_grid_metrics_MIDAS(xo, yo); oan = orc.angle_x(xo, yo)
rep("midas dx", dx, odx); rep("midas dy", dy, ody); rep("midas area", ar, oar); rep("angle", an, oan)
rng = np.random.default_rng(0)
xx = xo[:200,:333] + rng.normal(0,0.3,(200,333)); yy = yo[:200,:333] + rng.normal(0,0.3,(200,333))
nj1, ni1 = xx.shape
dx=np.empty((nj1,ni1-1)); dy=np.empty((nj1-1,ni1)); ar=np.empty((nj1-1,ni1-1)); an=np.empty((nj1,ni1))
for fix in (1,0):
    L.call("ogg_grid_metrics_midas", nj1, ni1, L.ptr(xx), L.ptr(yy), 6371.0e3, fix, L.ptr(dx), L.ptr(dy), L.ptr(ar))
    odx, ody, oar = orc.generate_grid_metrics_MIDAS(xx, yy, latlon_areafix=bool(fix))
    rep("distorted dx fix=%d"%fix, dx, odx); rep("distorted dy", dy, ody); rep("distorted area", ar, oar)
L.call("ogg_angle_x", nj1, ni1, L.ptr(xx), L.ptr(yy), L.ptr(an)); rep("distorted angle", an, orc.angle_x(xx, yy))
x2=np.empty((5,25)); y2=np.empty((5,25))
L.call("ogg_generate_latlon_grid", 24, 5, -300.0, 360.0, -78.0, 11.3, 1, L.ptr(x2), L.ptr(y2)); ox, oy = orc.generate_latlon_grid(24,5,-300.0,360,-78.0,11.3,True)
rep("latlon x", x2, ox); rep("latlon y", y2, oy)
# bipolar
Ni, Nj, lat0, lon_bp = 1440, 238, 64.97316302279852, -300.0
rp = np.tan(0.5*(90-lat0)*orc.PI_180)
ls=np.empty((Nj+1,Ni+1)); ps=np.empty((Nj+1,Ni+1)); hi=np.empty((Nj+1,Ni)); hj=np.empty((Nj,Ni+1))
L.call("ogg_bipolar_cap_mesh", Ni, Nj, lat0, lon_bp, L.ptr(ls), L.ptr(ps), L.ptr(hi), L.ptr(hj))
o = orc.generate_bipolar_cap_mesh(Ni, Nj, lat0, lon_bp, False)
for n,a,b in zip(("bp lams","bp phis","bp hi","bp hj"), (ls,ps,hi,hj), o): rep(n,a,b)
for order in (2,3,4,5):
    dxq=np.empty((Nj+1,Ni)); dyq=np.empty((Nj,Ni+1)); daq=np.empty((Nj,Ni))
    t=time.time(); L.call("ogg_bipolar_cap_metrics_quad", order, Ni, Nj, lat0, lon_bp, rp, 6371.0e3, L.ptr(dxq), L.ptr(dyq), L.ptr(daq)); t=time.time()-t
    o = orc.bipolar_cap_metrics_quad_fast(order, Ni, Nj, lat0, lon_bp, rp)
    for n,a,b in zip(("bpq%d dx"%order,"bpq%d dy"%order,"bpq%d area"%order), (dxq,dyq,daq), o): rep(n,a,b)
    print("   time incl. copies %.3fs" % t)
# displaced pole
for (Ni, Nj, r_dp) in ((72, 14, 0.2), (1440, 140, 0.2), (720, 70, 0.34135899793333113)):
    lon0, lat0sc, lon_dp = -300.0, -78.0, 80.0
    i = np.arange(Ni+1, dtype=np.float64); j = np.arange(Nj+1, dtype=np.float64)
    lam=np.empty((Nj+1,Ni+1)); phi=np.empty((Nj+1,Ni+1))
    L.call("ogg_displaced_pole_mesh", Ni+1, L.ptr(i), Nj+1, L.ptr(j), Ni, Nj, lon0, lat0sc, lon_dp, r_dp, L.ptr(lam), L.ptr(phi))
    o = orc.generate_displaced_pole_grid(Ni, Nj, lon0, lat0sc, lon_dp, r_dp)
    rep("dp x %d"%Ni, lam, o[0]); rep("dp y %d"%Ni, phi, o[1])
    for order in (2,4):
        dxq=np.empty((Nj+1,Ni)); dyq=np.empty((Nj,Ni+1)); daq=np.empty((Nj,Ni))
        t=time.time(); L.call("ogg_displaced_pole_metrics_quad", order, Ni, Nj, lon0, lat0sc, lon_dp, r_dp, 6371.0e3, L.ptr(dxq), L.ptr(dyq), L.ptr(daq)); t=time.time()-t
        o = orc.displacedPoleCap_metrics_quad(order, Ni, Nj, lon0, lat0sc, lon_dp, r_dp)
        for n,a,b in zip(("dpq%d dx"%order,"dpq%d dy"%order,"dpq%d area"%order), (dxq,dyq,daq), o): rep(n,a,b)
        print("   time incl. copies %.3fs" % t)
    fi = np.sort(rng.uniform(0, Ni+1, 300)); fj = np.sort(rng.uniform(-0.002, Nj+1, 9))
    for fd in (2,4,6):
        hi=np.empty((9,300)); hj=np.empty((9,300))
        L.call("ogg_displaced_pole_numerical_h", 300, L.ptr(fi), 9, L.ptr(fj), Ni, Nj, lon0, lat0sc, lon_dp, r_dp, 1e-3, fd, L.ptr(hi), L.ptr(hj))
        rep("num_hi fd%d"%fd, hi, orc.numerical_hi(fj, fi, Ni, Nj, lon0, lat0sc, lon_dp, r_dp, 1e-3, fd)); rep("num_hj fd%d"%fd, hj, orc.numerical_hj(fj, fi, Ni, Nj, lon0, lat0sc, lon_dp, r_dp, 1e-3, fd))
print("DONE")
