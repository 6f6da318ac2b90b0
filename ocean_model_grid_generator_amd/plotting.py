"""Visual checks behind --plotem and the two small row-selection helpers they use (the reference's OGG:604-679: cut_below, cut_above,
plot_mesh_in_latlon, plot_mesh_in_xyz, displacedPoleCap_plot).  Host-side only: numpy on arrays that are already on the host,
matplotlib imported when a plot is asked for (and a clear error if it is not installed).  Same names, arguments and defaults as the
reference, so that code which imports them from the drop-in module keeps working."""
import numpy as np

PI_180 = np.pi / 180.0


def _first_row_above(phi, lat):
    """Index of the first row whose first latitude exceeds `lat`; like the reference's loop, the LAST row when none does."""
    col = np.asarray(phi)[:, 0]
    above = np.nonzero(col > lat)[0]
    return int(above[0]) if above.size else col.shape[0] - 1


def cut_below(lam, phi, lowerlat):
    """Rows from the first one whose phi[:, 0] exceeds lowerlat (OGG:604-611)."""
    j = _first_row_above(phi, lowerlat)
    return lam[j:, :], phi[j:, :]


def cut_above(lam, phi, upperlat):
    """Rows before the first one whose phi[:, 0] exceeds upperlat (OGG:614-621)."""
    j = _first_row_above(phi, upperlat)
    return lam[0:j, :], phi[0:j, :]


def _pyplot():
    try:
        import matplotlib.pyplot as plt
    except ImportError as exc:   # the flag must not be ignored silently
        raise Exception("--plotem / the plot helpers need matplotlib, which is not installed: %s" % exc)
    return plt


def plot_mesh_in_latlon(lam, phi, stride=1, phi_color="k", lam_color="r", newfig=True, title=None, axis=None, block=False):
    """Every stride-th grid line of a mesh in the (lam, phi) plane (OGG:625-651)."""
    plt = _pyplot()
    if phi.shape != lam.shape:
        raise Exception("Ooops: lam and phi should have same shape")
    nj, ni = lam.shape
    if newfig:
        plt.figure(figsize=(10, 10))
    target = plt if axis is None else axis
    for i in range(0, ni, stride):
        target.plot(lam[:, i], phi[:, i], lam_color)
    for j in range(0, nj, stride):
        target.plot(lam[j, :], phi[j, :], phi_color)
    if title is not None:
        plt.title(title)
    if not block:
        plt.show()


def plot_mesh_in_xyz(lam, phi, stride=1, phi_color="k", lam_color="r", lowerlat=None, upperlat=None, newfig=True, title=None, axis=None,
                     block=False):
    """The same seen from above a pole: the mesh's Cartesian x, y on the unit sphere (OGG:654-664)."""
    if lowerlat is not None:
        lam, phi = cut_below(lam, phi, lowerlat=lowerlat)
    if upperlat is not None:
        lam, phi = cut_above(lam, phi, upperlat=upperlat)
    x = np.cos(phi * PI_180) * np.cos(lam * PI_180)
    y = np.cos(phi * PI_180) * np.sin(lam * PI_180)
    plot_mesh_in_latlon(x, y, stride=stride, phi_color=phi_color, lam_color=lam_color, newfig=newfig, title=title, axis=None, block=False)


def displacedPoleCap_plot(x_s, y_s, lon0, lon_dp, lat0, stride=40, block=False, dplat=None):
    """The southern cap on polar axes, the displaced pole marked (OGG:667-679).  Returns the axes."""
    plt = _pyplot()
    plt.figure(figsize=(10, 10))
    ax = plt.axes(projection="polar")
    plot_mesh_in_latlon(x_s, y_s, stride=stride, newfig=False, axis=ax, block=block)
    if dplat is not None:
        ax.plot(lon_dp, dplat, color="r", marker="*")
    return ax
