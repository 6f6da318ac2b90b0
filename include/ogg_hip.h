/* libogg_hip.so -- C ABI of the MI355X (gfx950) supergrid hot path.
 *
 * The reference (nikizadehgfdl/ocean_model_grid_generator, ocean_grid_generator.py, cited as OGG:<line>) has no
 * FFI: its boundary is the set of Python call sites in main() (OGG:1004-1170).  Each entry point below replaces
 * one of those callees; the Python host (ocean_model_grid_generator_amd/ocean_grid_generator.py) binds them
 * with ctypes under the reference's own function names.
 *
 * Conventions
 *   - all arrays are fp64, C order [j][i], i fastest, densely packed; sizes are given per argument;
 *   - entry points WITHOUT the _dev suffix take HOST pointers (caller-allocated, caller-owned; the library
 *     stages through device memory it allocates and frees inside the call);
 *   - entry points WITH the _dev suffix take DEVICE pointers and a hipStream_t (passed as void*), enqueue
 *     work and return without synchronising; no allocation, no host sync (graph-capturable);
 *   - every function returns OGG_OK or an error code; ogg_last_error() gives the text (thread-local);
 *   - no C++ exception crosses this boundary; calls may come from any thread, one call at a time per stream.
 */
#ifndef OGG_HIP_H
#define OGG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define OGG_OK 0
#define OGG_EORDER 1 /* "Uncoded order" (OGG:204,222,255) / "order not coded" (OGG:547,562) */
#define OGG_ESHAPE 2 /* "Input arrays do not have the same shape!" (OGG:722) and other size errors */
#define OGG_EHIP 3   /* a HIP runtime call failed (no device, launch failure, ...) */
#define OGG_ENOMEM 4 /* device allocation failed */
#define OGG_EARG 5   /* null pointer / invalid scalar */

#define OGG_DP_ARC_LITERAL 0 /* arc form of the displaced-pole quadrature: see ogg_displaced_pole_metrics_quad_form_ws_dev */
#define OGG_DP_ARC_CHORD 1

/* Mirror symmetry of the two caps (the `symmetry` field of ogg_bipolar_band / ogg_dpole_band, the *_sym_* entry points).
 *   OGG_SYM_MIRROR  evaluate the columns that determine the rest and write their mirror images: the bipolar projection (OGG:33-100) is
 *                   symmetric about its two pole meridians and its fold lines, so a quarter of the columns determines a row; the
 *                   displaced-pole map (OGG:447-467) about the meridian through lon_dp, so half of them does.  The columns where the
 *                   reference's own results are not mirror images of each other to within its own rounding error (next to the fold
 *                   lines, the pole meridians of the mesh, the rows next to the pole points) are still evaluated one by one: DESIGN.md.
 *   OGG_SYM_NONE    every column evaluated, as the reference does (OGG:168-172, 583-584).
 *   OGG_SYM_DEFAULT (0, what a zero-initialised descriptor asks for) the library's default: OGG_SYM_MIRROR unless the environment says
 *                   OGG_CAP_SYMMETRY=0 / none. */
#define OGG_SYM_DEFAULT 0
#define OGG_SYM_MIRROR 1
#define OGG_SYM_NONE 2

const char* ogg_last_error(void);
/* "ogg_hip <version> (gfx950) src <hash>": <hash> = first 12 hex digits of the sha256 over the kernel sources the library was built
 * from (csrc/build.py source_hash()); profiles/valu_counters.json and hbm_traffic.json record the hash they were collected with */
const char* ogg_version(void);
/* sizeof(ogg_latlon_band) for which = 0, sizeof(ogg_bipolar_band) for which = 1, sizeof(ogg_dpole_band) for which = 2 (-1
 * otherwise): lets a binding verify its layout */
long ogg_abi_sizeof(int which);
int ogg_device_count(int* count);
int ogg_set_device(int device);
int ogg_device_name(char* buf, int buflen);

/* ------------------------------------------------------------------------------------------------------
 * Mercator / regular lat-lon builders
 * ---------------------------------------------------------------------------------------------------- */

/* y_mercator_rounded (OGG:309-311, with y_mercator OGG:292-295): y* = sign(y)*round_half_even(|y|),
 * y = R*log((1+sin(phi))/cos(phi)), R = Ni/(2*pi).  phi in radians.  ystar: int64[n]. */
int ogg_y_mercator_rounded(long Ni, long n, const double* phi_rad, long long* ystar);
int ogg_y_mercator_rounded_dev(long Ni, long n, const double* phi_rad, long long* ystar, void* stream);

/* y_mercator (OGG:292-295): the unrounded ordinate y = R*log((1+sin(phi))/cos(phi)) */
int ogg_y_mercator(long Ni, long n, const double* phi_rad, double* y);
int ogg_y_mercator_dev(long Ni, long n, const double* phi_rad, double* y, void* stream);

/* phi_mercator (OGG:298-301): phi = atan(sinh(y/R))*(180/pi), degrees, for arbitrary ordinates y[n]. */
int ogg_phi_mercator(long Ni, long n, const double* y, double* phi_deg);
int ogg_phi_mercator_dev(long Ni, long n, const double* y, double* phi_deg, void* stream);
/* same for the integer ordinates y0, y0+1, ..., y0+n-1 (the axis of OGG:336), no input array */
int ogg_mercator_axis_dev(long Ni, long long y0, long n, double* phi_deg, void* stream);

/* out[k] = a0 + (k*len)/denom, k = 0..n-1: the axes of OGG:113,115,431,834,835 */
int ogg_linear_axis_dev(long n, double a0, double len, double denom, double* out, void* stream);

/* out[k] = a0 + (idx[k]*len)/denom for arbitrary (fractional) indices idx[n]: OGG:126-127, 479-482 */
int ogg_affine_index(long n, const double* idx, double a0, double len, double denom, double* out);
int ogg_affine_index_dev(long n, const double* idx, double a0, double len, double denom, double* out, void* stream);

/* np.tile pair (OGG:430-432, OGG:840-841): x[j][i] = lon1d[i], y[j][i] = lat1d[j]; rows j0..j0+nrows-1 of the
 * lat axis are written to nrows x ni1 outputs. */
int ogg_tile_latlon(long nrows, long ni1, const double* lat1d, const double* lon1d, double* x, double* y);
int ogg_tile_latlon_dev(long nrows, long ni1, const double* lat1d, const double* lon1d, double* x, double* y,
                        void* stream);

/* generate_latlon_grid (OGG:832-846) incl. both axes; skip_first_row=1 reproduces the ensure_nj_even row drop
 * (OGG:836-838).  x, y: (lnj+1-skip_first_row) x (lni+1). */
int ogg_generate_latlon_grid(long lni, long lnj, double llon0, double llen_lon, double llat0, double llen_lat,
                             int skip_first_row, double* x, double* y);

/* ------------------------------------------------------------------------------------------------------
 * MIDAS stencil metrics (OGG:687-716) and grid orientation angle (OGG:719-729), fused
 * ---------------------------------------------------------------------------------------------------- */

/* mdist (OGG:682-684): min(mod(x1-x2,360), mod(x2-x1,360)) with numpy's sign-of-divisor mod, element-wise */
int ogg_mdist(long n, const double* x1, const double* x2, double* out);
int ogg_mdist_dev(long n, const double* x1, const double* x2, double* out, void* stream);

/* x, y: nrows_xy x ni1 point rows.  Writes
 *   dx    [n_pt_rows  ][ni1-1]   (rows 0..n_pt_rows-1)            if dx    != NULL
 *   angle [n_pt_rows  ][ni1  ]                                     if angle != NULL
 *   dy    [n_cell_rows][ni1  ]   (needs point row j+1: n_cell_rows+1 <= nrows_xy)   if dy != NULL
 *   area  [n_cell_rows][ni1-1]                                     if area  != NULL
 * A full sub-grid has n_pt_rows = nrows_xy = nj+1, n_cell_rows = nj.  A latitude band passes its own rows plus one
 * halo row (the first row of the band above) and n_pt_rows = n_cell_rows = nrows_xy-1. */
int ogg_grid_metrics_midas_dev(long nrows_xy, long ni1, const double* x, const double* y, long n_pt_rows,
                               long n_cell_rows, double Re, int latlon_areafix, double* dx, double* dy,
                               double* area, double* angle, void* stream);
/* generate_grid_metrics_MIDAS(x, y, Re, latlon_areafix): dx (nj1 x ni1-1), dy (nj1-1 x ni1), area (nj1-1 x ni1-1) */
int ogg_grid_metrics_midas(long nj1, long ni1, const double* x, const double* y, double Re, int latlon_areafix,
                           double* dx, double* dy, double* area);
/* angle_x(x, y): angle_dx (nj1 x ni1), degrees */
int ogg_angle_x(long nj1, long ni1, const double* x, const double* y, double* angle_dx);

/* Fused builder for sub-grids that are lat-lon by construction (x[j][i] = lon1d[i], y[j][i] = lat1d[j]): writes x, y,
 * angle_dx (n_pt_rows x ni1), dx (n_pt_rows x ni1-1) and dy (n_cell_rows x ni1), area (n_cell_rows x ni1-1) from the two
 * 1-D axes alone; bit-identical to ogg_tile_latlon_dev + ogg_grid_metrics_midas_dev.  lat1d must hold n_pt_rows entries,
 * plus one more when n_cell_rows == n_pt_rows (the row above the band).  metrics = 0 writes only x, y, angle_dx. */
typedef struct {
    int axis_kind;  /* 0: lat[k] = a0 + (k*len)/denom (OGG:835); 1: Mercator, lat[k] = atan(sinh((y0+k)/R))*(180/pi), R = Ni/(2 pi)
                       (OGG:336); 2: lat[k] = lat1d[k] (device array, e.g. the enhanced-equator axis) */
    double a0, len, denom;
    long long y0;
    const double* lat1d;
    long k0;           /* axis index of the band's first point row */
    long n_pt_rows;    /* point rows of the band (x, y, dx, angle_dx) */
    long n_cell_rows;  /* cell rows of the band (dy, area); row n_cell_rows of the axis must exist when == n_pt_rows */
    double *x, *y, *dx, *dy, *area, *angle;
} ogg_latlon_band;
/* Up to 4 lat-lon bands in one launch, axes evaluated in the kernel: lon[i] = lon0 + (i*lenlon)/(ni1-1) (OGG:431, 834). */
int ogg_latlon_supergrid_multi_dev(int n_bands, const ogg_latlon_band* bands, long ni1, double lon0, double lenlon, double Re,
                                   int metrics, void* stream);
/* The same fields, the same bits, for launches that carry nothing but lat-lon sub-grids: the unit of work is one ROW of one FIELD and
 * units are taken in (band, field, row) order, so the chip writes a compact window that sweeps through one array at a time -- the
 * pattern its HBM write path rewards -- fed from a row table and a column table built by a first small launch in the caller's
 * workspace (>= ogg_latlon_rows_workspace_bytes bytes of device memory).  Two launches, no allocation. */
long ogg_latlon_rows_workspace_bytes(int n_bands, const ogg_latlon_band* bands, long ni1);
int ogg_latlon_supergrid_rows_ws_dev(int n_bands, const ogg_latlon_band* bands, long ni1, double lon0, double lenlon, double Re,
                                     int metrics, void* workspace, long workspace_bytes, void* stream);
int ogg_latlon_supergrid_dev(long n_pt_rows, long n_cell_rows, long ni1, const double* lat1d, const double* lon1d, double Re,
                             int metrics, double* x, double* y, double* dx, double* dy, double* area, double* angle,
                             void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Murray bipolar Arctic cap
 * ---------------------------------------------------------------------------------------------------- */

/* bipolar_projection (OGG:33-100), element-wise over n points.  metrics_only != 0: lams, phis may be NULL. */
int ogg_bipolar_projection(long n, const double* lamg, const double* phig, double lon_bp, double rp,
                           int metrics_only, double* lams, double* phis, double* h_i_inv, double* h_j_inv);
int ogg_bipolar_projection_dev(long n, const double* lamg, const double* phig, double lon_bp, double rp,
                               int metrics_only, double* lams, double* phis, double* h_i_inv, double* h_j_inv,
                               void* stream);

/* generate_bipolar_cap_mesh (OGG:103-122), rows j0..j0+nrows-1 of the (Nj+1) x (Ni+1) mesh.
 * lams, phis: nrows x (Ni+1).  h_i_inv: nrows x Ni and h_j_inv: nrows x (Ni+1), already scaled as OGG:119-120
 * (h_j_inv row Nj is computed but dropped by the reference; rows >= Nj are not written); either may be NULL. */
int ogg_bipolar_cap_mesh_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, double* lams,
                             double* phis, double* h_i_inv, double* h_j_inv, void* stream);
/* same, also writing angle_dx = angle_x(lams, phis) (nrows x (Ni+1); NULL to skip) without reading the mesh back */
int ogg_bipolar_cap_mesh_angle_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, double* lams,
                                   double* phis, double* h_i_inv, double* h_j_inv, double* angle_dx, void* stream);
/* same with the column symmetry stated by the caller (OGG_SYM_*; the forms above ask for OGG_SYM_DEFAULT): mirrored, the columns
 * [0, Ni/4] (+ 2 degrees beyond the pole meridian) are evaluated and written to their three images, the columns next to the fold lines
 * and within 2 degrees of the pole meridians at their own positions */
int ogg_bipolar_cap_mesh_angle_sym_dev(long Ni, long Nj, double lat0_bp, double lon_bp, long j0, long nrows, int symmetry, double* lams,
                                       double* phis, double* h_i_inv, double* h_j_inv, double* angle_dx, void* stream);
int ogg_bipolar_cap_mesh(long Ni, long Nj, double lat0_bp, double lon_bp, double* lams, double* phis,
                         double* h_i_inv, double* h_j_inv);
int ogg_bipolar_cap_mesh_sym(long Ni, long Nj, double lat0_bp, double lon_bp, int symmetry, double* lams, double* phis,
                             double* h_i_inv, double* h_j_inv);

/* bipolar_cap_ij_array (OGG:125-133): per-index arc lengths (radians) at fractional indices i[n_i], j[n_j];
 * h_i_inv, h_j_inv: n_j x n_i */
int ogg_bipolar_cap_ij_array(long n_i, const double* i, long n_j, const double* j, long Ni, long Nj, double lat0_bp,
                             double lon_bp, double rp, double* h_i_inv, double* h_j_inv);
int ogg_bipolar_cap_ij_array_dev(long n_i, const double* i, long n_j, const double* j, long Ni, long Nj, double lat0_bp,
                                 double lon_bp, double rp, double* h_i_inv, double* h_j_inv, void* stream);

/* bipolar_cap_metrics_quad_fast (OGG:136-188): Gauss-Lobatto quadrature (order 2..5) of the analytic scale
 * factors.  Band form: dxq rows j0..j0+n_dx_rows-1 of (ny+1) x nx; dyq rows j0..j0+n_cell_rows-1 of ny x (nx+1);
 * daq rows j0..j0+n_cell_rows-1 of ny x nx. */
int ogg_bipolar_cap_metrics_quad_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp,
                                     double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq,
                                     double* daq, void* stream);
int ogg_bipolar_cap_metrics_quad(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                 double* dxq, double* dyq, double* daq);
int ogg_bipolar_cap_metrics_quad_sym(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re, int symmetry,
                                     double* dxq, double* dyq, double* daq);
/* Same with a caller-provided device workspace for the row/column tables (at least ogg_bipolar_quad_workspace_bytes
 * bytes): no allocation of any kind inside the call, so it can be captured into a HIP graph.  The plain _dev form takes
 * the tables from the stream-ordered allocator (hipMallocAsync / hipFreeAsync). */
long ogg_bipolar_quad_workspace_bytes(int order, long nx, long ny);
int ogg_bipolar_cap_metrics_quad_ws_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                        long j0, long n_dx_rows, long n_cell_rows, double* dxq, double* dyq, double* daq,
                                        void* workspace, long workspace_bytes, void* stream);
/* Same with the column symmetry stated by the caller (OGG_SYM_*, top of this file); the forms above ask for OGG_SYM_DEFAULT.  Mirrored,
 * the rows below 88.2 degrees are evaluated on the cells [0, nx/4) and next to the fold lines (6 degrees either side of the columns
 * 0 and nx/2) and written to their images; replaces the column loop of OGG:168-172 for those rows. */
int ogg_bipolar_cap_metrics_quad_sym_ws_dev(int order, long nx, long ny, double lat0_bp, double lon_bp, double rp, double Re,
                                            long j0, long n_dx_rows, long n_cell_rows, int symmetry, double* dxq, double* dyq, double* daq,
                                            void* workspace, long workspace_bytes, void* stream);

/* One rank's share of a tripolar supergrid -- the lat-lon sub-grids and the bipolar cap generated by the sub-grid loop of
 * main() (OGG:1100-1313: generate_mercator_grid / generate_latlon_grid / generate_bipolar_cap_mesh + angle_x +
 * bipolar_cap_metrics_quad_fast) -- in three launches on one stream: the HBM-bound lat-lon row strips and the VALU-bound
 * cap workgroups share each launch, so nothing waits on a cross-stream dependency.  Results are bit-identical to
 * ogg_latlon_supergrid_multi_dev + ogg_bipolar_cap_mesh_angle_dev + ogg_bipolar_cap_metrics_quad_ws_dev on the same bands.
 * cap may be NULL (no cap rows on this rank); metrics == 0 writes coordinates and angle_dx only. */
typedef struct ogg_bipolar_band {
    long Ni, Nj;             /* the cap is (Nj+1) x (Ni+1) points */
    double lat0_bp, lon_bp;  /* OGG:103 */
    double rp, Re;           /* OGG:117; sphere radius */
    int order;               /* Gauss-Lobatto order of the quadrature, 2..5 (OGG:191-204) */
    int symmetry;            /* OGG_SYM_DEFAULT (0), OGG_SYM_MIRROR or OGG_SYM_NONE: mesh and quadrature from a quarter of the columns */
    long j0;                 /* first mesh row of the band */
    long n_pt_rows;          /* point rows: x, y, angle (n_pt_rows x (Ni+1)), dx (n_pt_rows x Ni) */
    long n_cell_rows;        /* cell rows: dy (n_cell_rows x (Ni+1)), area (n_cell_rows x Ni); n_pt_rows - 1 on the band that
                                holds row Nj, else n_pt_rows */
    double *x, *y, *angle, *dx, *dy, *area;
    void* workspace;         /* >= ogg_bipolar_quad_workspace_bytes(order, Ni, Nj) bytes of device memory (metrics only): tables,
                                fix-up list, claim counters of the lat-lon strips; one pass at a time per workspace */
    long workspace_bytes;
} ogg_bipolar_band;
int ogg_tripolar_pass_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re,
                          int metrics, const ogg_bipolar_band* cap, void* stream);
/* same; events4 (NULL, or 4 events from ogg_event_create) are recorded on the stream before the first launch and after each
 * of the three launches, so that the caller can time them (ogg_event_elapsed_ms); alg_bytes3 (NULL, or 3 doubles on the host)
 * receives the bytes each launch writes (its share of the 48 B per cell) */
int ogg_tripolar_pass_events_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re,
                                 int metrics, const ogg_bipolar_band* cap, void** events4, double* alg_bytes3, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Displaced-pole Southern cap
 * ---------------------------------------------------------------------------------------------------- */

/* displacedPoleCap_mesh (OGG:488-506) at index vectors i[n_i], j[n_j] (may be fractional), including the
 * sequential 360-degree unwrap along i (monotonic_bounding, OGG:470-475).  lams, phis: n_j x n_i. */
int ogg_displaced_pole_mesh_dev(long n_i, const double* i, long n_j, const double* j, long ni, long nj, double lon0,
                                double lat0, double lam_pole, double r_pole, double* lams, double* phis,
                                void* stream);
int ogg_displaced_pole_mesh(long n_i, const double* i, long n_j, const double* j, long ni, long nj, double lon0,
                            double lat0, double lam_pole, double r_pole, double* lams, double* phis);
/* displacedPoleCap_projection (OGG:447-467) on explicit lon/lat grids (nj x ni) with z_0 = z0_re + i z0_im and
 * r_joint given; x_0 is the seed of the unwrap (the reference passes lon_grid[0,0], OGG:463). */
int ogg_displaced_pole_projection(long nj, long ni, const double* lon_grid, const double* lat_grid, double z0_re,
                                  double z0_im, double r_joint, double x_0, double* lam, double* phi);
int ogg_displaced_pole_projection_dev(long nj, long ni, const double* lon_grid, const double* lat_grid, double z0_re,
                                      double z0_im, double r_joint, double x_0, double* lam, double* phi, void* stream);
/* monotonic_bounding (OGG:470-475), in place on x (nj x ni) */
int ogg_monotonic_bounding(long nj, long ni, double* x, double x_0);
int ogg_monotonic_bounding_dev(long nj, long ni, double* x, double x_0, void* stream);
/* the haversine of great_arc_distance (OGG:527-532), element-wise; inputs in degrees, output in radians */
int ogg_haversine(long n, const double* lam0, const double* phi0, const double* lam1, const double* phi1, double* out);
int ogg_haversine_dev(long n, const double* lam0, const double* phi0, const double* lam1, const double* phi1, double* out,
                      void* stream);

/* generate_displaced_pole_grid (OGG:509-518), rows j0..j0+nrows-1 of (Nj+1) x (Ni+1): projection, the sequential unwrap of
 * monotonic_bounding (a look-back scan over the column strips of a row) and, when angle_dx != NULL, angle_x (OGG:719-729) of
 * the same rows (nrows x (Ni+1)) without reading the mesh back -- one launch.  The _ws form takes a caller-provided device
 * workspace (>= ogg_displaced_pole_grid_workspace_bytes; its first two 32-bit words are the work counter and an error flag,
 * see ogg_workspace_error_flag_dev) and allocates nothing; the plain form uses the stream-ordered allocator. */
int ogg_displaced_pole_grid_dev(long Ni, long Nj, double lon0, double lat0, double lon_dp, double r_dp, long j0,
                                long nrows, double* x, double* y, void* stream);
long ogg_displaced_pole_grid_workspace_bytes(long Ni, long nrows);
int ogg_displaced_pole_grid_angle_ws_dev(long Ni, long Nj, double lon0, double lat0, double lon_dp, double r_dp, long j0,
                                         long nrows, double* x, double* y, double* angle_dx, void* workspace,
                                         long workspace_bytes, void* stream);
/* Error flag of a workspace used by a displaced-pole call (*flag != 0: a look-back wait gave up, the results of that call are
 * invalid; never observed, the spin is bounded so that a grid always drains).  Synchronises the stream. */
int ogg_workspace_error_flag_dev(const void* workspace, int* flag, void* stream);

/* numerical_hi / numerical_hj (OGG:535-562; great_arc_distance OGG:522-532) on the lattice j[n_j] x i[n_i].
 * fd_order in {2,4,6}.  h_i, h_j: n_j x n_i (either may be NULL). */
int ogg_displaced_pole_numerical_h_dev(long n_i, const double* i, long n_j, const double* j, long nx, long ny,
                                       double lon0, double lat0, double lon_dp, double r_dp, double eps,
                                       int fd_order, double* h_i, double* h_j, void* stream);
int ogg_displaced_pole_numerical_h(long n_i, const double* i, long n_j, const double* j, long nx, long ny,
                                   double lon0, double lat0, double lon_dp, double r_dp, double eps, int fd_order,
                                   double* h_i, double* h_j);

/* displacedPoleCap_metrics_quad (OGG:565-601): quadrature order (2..5) is also the finite-difference order, as in
 * the reference (OGG:583-584), so only orders 2 and 4 are valid.  Band form as for the bipolar cap; cell rows
 * below j0 are simply not evaluated (main() discards the doughnut rows, OGG:1177-1186).
 * arc_form selects how the great-arc distance between two probes of the finite-difference stencil is taken:
 *   OGG_DP_ARC_CHORD    the finite-difference stencil of the reference, distance of two probes from their positions on the sphere (no
 *                       atan2, no unwrap): ~6x less arithmetic than the literal form, and closer to what the reference's formula means
 *                       (below).  Since round 4 THE default: what the entry points without an arc_form argument run, and what the
 *                       Python host (main(), displacedPoleCap_metrics_quad, SupergridPlan, bench.py) asks for unless OGG_DP_ARC /
 *                       dp_arc / arc_form says literal;
 *   OGG_DP_ARC_LITERAL  the reference's operation sequence (haversine of the projected, unwrapped longitudes and latitudes,
 *                       OGG:522-532): opt-in through the *_form entry points and the band descriptor of the pass.
 * Measured on the 5760 x 560 cap of BASELINE config 4 (1/8 degree), max relative difference of dx / dy / area
 *   from the numpy oracle (profiles/dp_parity.json):             literal 1.3e-9 / 1.2e-9 / 7.6e-10   chord 1.5e-9 / 1.3e-9 / 9.8e-10
 *   from a 50-digit evaluation of the reference's own formula on 10 500 cells (tests/golden/truth_table.npz, tests/test_gpu_truth.py,
 *   profiles/r04_truth_table.json):   the fp64 reference itself 1.4e-9 / 9.0e-10 / 1.2e-9   literal 1.4e-9 / 8.5e-10 / 1.2e-9
 *                                     chord 7.6e-10 / 2.7e-10 / 8.3e-10
 * Both grow with the resolution (7e-12 at Ni = 72): the reference differentiates an arc of 2e-3 index units numerically, so ONE ulp of
 * atan2 is amplified by ~Ni / (4e-3 pi); the fp64 reference is itself that far from the exact value of its formula, and the chord form,
 * which never forms a longitude, is the closest of the three.  The literal form is not bit-identical to a CPU run either, because the
 * transcendental functions are not (the restatements of atan / atan2 used here reproduce the bits of the ROCm 7.2 device library,
 * ogg_libm_check_dev). */
/* (OGG_DP_ARC_LITERAL = 0, OGG_DP_ARC_CHORD = 1: defined next to the error codes at the top of this file) */
int ogg_displaced_pole_metrics_quad_dev(int order, long nx, long ny, double lon0, double lat0, double lon_dp,
                                        double r_dp, double Re, long j0, long n_dx_rows, long n_cell_rows,
                                        double* dxq, double* dyq, double* daq, void* stream);
/* Same with a caller-provided workspace for the row / column tables and the look-back words (at least
 * ogg_displaced_pole_quad_workspace_bytes(order, nx, n_cell_rows) bytes; a few MB -- the lattice itself is never stored): no
 * allocation inside the call.  One call at a time per workspace. */
long ogg_displaced_pole_quad_workspace_bytes(int order, long nx, long n_cell_rows);
int ogg_displaced_pole_metrics_quad_ws_dev(int order, long nx, long ny, double lon0, double lat0, double lon_dp,
                                           double r_dp, double Re, long j0, long n_dx_rows, long n_cell_rows, double* dxq,
                                           double* dyq, double* daq, void* workspace, long workspace_bytes, void* stream);
int ogg_displaced_pole_metrics_quad_form_ws_dev(int arc_form, int order, long nx, long ny, double lon0, double lat0,
                                                double lon_dp, double r_dp, double Re, long j0, long n_dx_rows,
                                                long n_cell_rows, double* dxq, double* dyq, double* daq, void* workspace,
                                                long workspace_bytes, void* stream);
/* same with the column symmetry stated by the caller (OGG_SYM_*; the form above asks for OGG_SYM_DEFAULT).  Mirrored (chord form only, when
 * the meridian of the displaced pole is a node column, (lon_dp - lon0) nx / 360 an integer, and nx is even): the half of the columns on
 * one side of that meridian is evaluated and written to its mirror images (OGG:583-584 evaluate every column) */
int ogg_displaced_pole_metrics_quad_form_sym_ws_dev(int arc_form, int symmetry, int order, long nx, long ny, double lon0, double lat0,
                                                    double lon_dp, double r_dp, double Re, long j0, long n_dx_rows, long n_cell_rows,
                                                    double* dxq, double* dyq, double* daq, void* workspace, long workspace_bytes,
                                                    void* stream);
int ogg_displaced_pole_metrics_quad(int order, long nx, long ny, double lon0, double lat0, double lon_dp, double r_dp,
                                    double Re, double* dxq, double* dyq, double* daq);
int ogg_displaced_pole_metrics_quad_form(int arc_form, int order, long nx, long ny, double lon0, double lat0, double lon_dp,
                                         double r_dp, double Re, double* dxq, double* dyq, double* daq);
int ogg_displaced_pole_metrics_quad_form_sym(int arc_form, int symmetry, int order, long nx, long ny, double lon0, double lat0, double lon_dp,
                                             double r_dp, double Re, double* dxq, double* dyq, double* daq);

/* One rank's share of a whole supergrid: ogg_tripolar_pass_dev plus the band of a displaced-pole southern cap (the SC branch of
 * the same sub-grid loop, OGG:1158-1197: generate_displaced_pole_grid + angle_x + displacedPoleCap_metrics_quad) in the SAME
 * launches: its tables and the reset of its look-back words join launch A, its mesh + angle workgroups and -- in the chord form --
 * its quadrature strips join launch B; the literal form of the quadrature is a fourth launch (D) on the same stream.  Results are
 * bit-identical to ogg_displaced_pole_grid_angle_ws_dev + ogg_displaced_pole_metrics_quad_form_ws_dev on the same band.
 * south_cap may be NULL.  events5 (NULL, or 5 events, entries may be NULL): before A, after A, B, C, D; alg_bytes4: bytes each
 * launch writes. */
typedef struct ogg_dpole_band {
    long Ni, Nj;                 /* the cap is (Nj+1) x (Ni+1) points (OGG:509) */
    double lon0, lat0;           /* OGG:478: first longitude, latitude of the joint with the Southern Ocean sub-grid */
    double lon_dp, r_dp;         /* OGG:495: longitude and radius of the displaced pole */
    double Re;
    int order;                   /* Gauss-Lobatto order of the quadrature = finite-difference order: 2 or 4 (OGG:583-584) */
    int arc_form;                /* OGG_DP_ARC_LITERAL or OGG_DP_ARC_CHORD */
    long j0;                     /* first mesh row of the band (rows below the doughnut cut are simply never asked for) */
    long n_pt_rows;              /* point rows: x, y, angle (n_pt_rows x (Ni+1)), dx (n_pt_rows x Ni) */
    long n_cell_rows;            /* cell rows: dy (n_cell_rows x (Ni+1)), area (n_cell_rows x Ni); n_pt_rows - 1 on the band that holds
                                    row Nj, else n_pt_rows */
    double *x, *y, *angle, *dx, *dy, *area;
    void* workspace;             /* >= ogg_dpole_band_workspace_bytes(order, Ni, n_pt_rows) bytes of device memory */
    long workspace_bytes;
    int symmetry;                /* OGG_SYM_DEFAULT (0), OGG_SYM_MIRROR or OGG_SYM_NONE: the chord-form quadrature from half of the columns */
} ogg_dpole_band;
long ogg_dpole_band_workspace_bytes(int order, long Ni, long n_pt_rows);
/* Host-side check of the mirrored kernels' column spaces (no GPU): replays the index arithmetic the kernels use for one row of a cap with n
 * cells -- which = 0 the bipolar quadrature, 1 the bipolar mesh, 2 the displaced-pole quadrature in the chord form (order 2 or 4; lon0,
 * lon_dp: the meridian of the displaced pole) -- and counts how often every cell (cell_writes[n]; NULL for which = 1) and every node column
 * (col_writes[n + 1]) is written, and how many are evaluated.  Every count must be 1 whatever `symmetry` (OGG_SYM_*). */
int ogg_symmetry_coverage(int which, int order, long n, double lon0, double lon_dp, int symmetry, int* cell_writes, int* col_writes,
                          long* evaluated);
int ogg_supergrid_pass_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re, int metrics,
                           const ogg_bipolar_band* cap, const ogg_dpole_band* south_cap, void** events5, double* alg_bytes4,
                           void* stream);
/* The same in two steps, for a caller that runs the pass of one set of bands many times (a rank's step loop): the plan holds the kernel
 * parameters of the launches, their grid sizes and the tiling knobs (the OGG_* environment variables are read when the plan is BUILT), so
 * that a run costs the host its launches and nothing else -- no validation, no planning, no getenv.  The band descriptors are copied:
 * they may be freed after ogg_supergrid_pass_plan_dev returns; the buffers and workspaces they point to must stay.  The runs of a plan
 * go to ONE stream at a time (switch streams only after synchronising the old one) and come from one host thread.
 * ogg_supergrid_pass_dev == plan + run + destroy, without what follows.
 *
 * The tables of the next pass ride in launch B.  With metrics (or a displaced-pole cap) the first launch of a pass writes only the cap
 * workspaces -- tables, cleared look-back words and counters -- and what it writes does not depend on the pass before it.  A plan
 * therefore owns TWO workspaces per cap (device allocations of its own, of the size of the caller's -- their contents outlive a run,
 * so the caller's workspace, which other entry points may use between two runs, is left alone; OGG_PASS_SLOTS=1, read when the plan is
 * built, turns this off: one slot, the caller's) and the last workgroups of launch B of one pass do launch A's work for
 * the next pass in the other slot; that pass then starts with launch B.  One packet less per pass on the stream (3-4.5 us), no second
 * stream, no flag and no wait: the stream's order is the dependence.  Every pass still builds one set of tables; the first pass of a
 * plan, and a pass that records events (events5 != NULL, so that they time it), run launch A themselves; the tables the LAST pass of a
 * plan built for a successor that never came are wasted (a few microseconds).  What the caller sees is unchanged: the outputs of a run
 * are complete when `stream` has executed it (a pass captured into a HIP graph: when a replay has executed; such a graph must not be
 * replayed after ogg_supergrid_pass_plan_destroy, which then waits for the whole device rather than for the last stream), and nothing of
 * pass k + 1 reaches an output array during pass k (the one output launch
 * A writes, the j = ny row of the bipolar dx, goes through the workspace and is copied by the tail launch); results are bit-identical
 * with one slot or two.  ogg_supergrid_pass_plan_destroy waits for the device. */
int ogg_supergrid_pass_plan_dev(int n_latlon, const ogg_latlon_band* latlon, long ni1, double lon0, double lenlon, double Re, int metrics,
                                const ogg_bipolar_band* cap, const ogg_dpole_band* south_cap, void** plan_out);
int ogg_supergrid_pass_run_dev(void* plan, void** events5, double* alg_bytes4, void* stream);
int ogg_supergrid_pass_plan_destroy(void* plan);
long ogg_supergrid_pass_plan_slots(const void* plan);          /* workspace slots of the plan: 2, or 1 (every pass runs its own launch A) */
long ogg_supergrid_pass_plan_carried_runs(const void* plan);   /* runs so far that started with launch B */
/* Waits for `stream`, then *flags = bit 1 / bit 2: a look-back wait of the displaced-pole mesh / quadrature timed out, in either slot, in
 * a pass since the slot's words were last cleared.  Results of such a pass are invalid; never observed. */
int ogg_supergrid_pass_plan_flags_dev(const void* plan, int* flags, void* stream);

/* ------------------------------------------------------------------------------------------------------
 * Device utilities used by the band-sharded pipeline (bench / multi-GPU)
 * ---------------------------------------------------------------------------------------------------- */
int ogg_fill_dev(long n, double value, double* out, void* stream);
/* Byte-swap-on-copy for write_nc's big-endian output (NetCDF classic, OGG:773-829): dst[k] = bswap64(src[k]), k < n.  src is device
 * memory; dst is device memory or pinned host memory (hipHostMalloc / torch pin_memory), into which the kernel stores directly. */
int ogg_bswap64_dev(long n, const void* src, void* dst, void* stream);
/* Self-test of the device-library functions the kernels restate with their coefficients as scalar operands (ogg_math.h,
 * ogg_bipolar_dev.h): which = 0: asin on [0, 1] (x); 1: atan (x, any); 2: atan2(y, x), finite; 3: 1.0 / x and 4: sqrt(x) without
 * scaling and special cases, 2^-700 <= x <= 2^700; 5: y / x without scaling, 2^-300 <= |x|, |y| <= 2^300 or y = +-0; 6: atan (x) and
 * 7: atan2(y, x) with their coefficients in vector registers (the literal displaced-pole quadrature's forms); 9: atan2(y, x) for ANY
 * arguments, infinities and NaNs included (the generic stencil kernel's form; two NaNs count as equal); 13 / 14: mdist(x, y) (OGG:682-684) from
 * one reduction (the generic stencil kernel's form) / from two, against numpy.mod's own fmod form.  (8, 10, 11, 12 -- restatements that no kernel uses any more -- were removed with them in round 4: OGG_EARG.)  The number of k < n for which the
 * restatement differs IN ANY BIT from the library's own function is ADDED to *n_diff (device memory, 8 bytes, zeroed by the caller). */
int ogg_libm_check_dev(int which, long n, const double* x, const double* y, unsigned long long* n_diff, void* stream);
/* The five sums behind metrics_error (OGG:732-770) of one sub-grid band, on the device and deterministic:
 * out5 = { sum(area), sum(dy[:, col_a]), sum(dy[:, col_b]) (0 when col_b < 0), sum(dx[0, :]) if want_first_row,
 * sum(dx[n_dx_rows-1, :]) if want_last_row }.  dx: n_dx_rows x ni, dy: n_cell_rows x (ni+1), area: n_cell_rows x ni; out5 is a
 * device pointer.  A band-sharded run adds the out5 of all ranks (one all-reduce) and evaluates OGG:735-770 on the host. */
int ogg_metrics_sums_dev(long n_dx_rows, long n_cell_rows, long ni, const double* dx, const double* dy, const double* area,
                         long col_a, long col_b, int want_first_row, int want_last_row, double* out5, void* stream);
/* per-launch timing of the dominant kernels with HIP events on the given stream: start/stop bracket */
int ogg_event_create(void** ev);
int ogg_event_destroy(void* ev);
int ogg_event_record(void* ev, void* stream);
int ogg_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms);
int ogg_stream_synchronize(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OGG_HIP_H */
