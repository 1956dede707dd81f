"""Cross-check of the restated cv::minAreaRect (oracle/mot_oracle_mar.c, "parity unpinned": OpenCV is in neither /root/reference nor this
image) against an INDEPENDENT exhaustive oracle (oracle/mot_oracle_mar_brute.c: monotone-chain hull, exact integer area of the enclosing
rectangle on every hull edge). TEST INFRASTRUCTURE. The reference's call: OT/src/cluster/box_fitting.cpp:357-362 (minAreaRect,
RotatedRect::points), consumed by getPointsInPcFrame :75-95.

check(pixels) asserts, for one point set:
  area        the restated rectangle's area (width x height) equals the exact minimum over all hull-edge-aligned enclosing rectangles to
              the rounding of the float32 caliper arithmetic OpenCV prescribes: width and height are float32 projections of hull-vertex
              differences (<= the hull's diameter D) on float32 unit vectors, so each carries an absolute error of a few eps32 x D and
              the area one of a few eps32 x D x (w + h). Measured over 12 000 sets: <= 3.3 such units (2.9e-7 relative at the 90th
              percentile, 6.8e-6 worst — thin rectangles, where a side is the small difference of two large projections). Bar:
              AREA_UNITS = 6 of those units (1e-6 relative would demand more than float32 calipers deliver on thin clusters).
              THIN HULLS: the walk decides which caliper advances by comparing COSINES in float32, and 1 - cos resolves angles only down to
              sqrt(2 eps32) = 3.5e-4 rad (0.02 degrees). On a hull whose consecutive edges differ by less than that — a 12 m wall, 360 x 1.3
              pixels — the float32 walk advances the wrong caliper now and then and ends on a rectangle whose direction is off by up to that
              angle: its thin side is then off by up to D x 3.5e-4 (0.12 pixel = 7 mm at D = 360) and no longer contains every point. That is
              the published algorithm in its prescribed precision, not a transcription slip: the SAME walk carried out in double
              (orc_min_area_rect_f64_area, instantiated from the same source text) equals the exhaustive minimum on every set, thin or not —
              asserted here for every set. Such sets (69 of 8 400 rectangle clusters of the rendered 154-frame streams, all walls thinner
              than 2 pixels, which the rule filter's 0.2 m minimum width rejects whatever their rectangle) pass under the second clause:
              both sides within D x sqrt(2 eps32) of the exact rectangle's; they are counted separately.
  alignment   one side of the restated rectangle is collinear with a hull edge whose own enclosing rectangle attains that minimum
  containment every point lies inside the restated rectangle (to float32 rounding)
  conventions OpenCV 3.2's: angle in [-90, 0] degrees for a proper rectangle; RotatedRect::points() = bottomLeft, topLeft, topRight,
              bottomRight of the rectangle's own frame (y down): pt0 = c - w/2 u + h/2 v, pt1 = c - w/2 u - h/2 v, pt2 = 2c - pt0,
              pt3 = 2c - pt1 with u = (cos a, sin a), v = (-sin a, cos a)
and returns (area error in units of eps32 x D x (w + h), number of exact ties, clause) with clause "rounding" or "cosine resolution"."""
import ctypes as C
import numpy as np

AREA_UNITS = 6.0    # x eps32 x hull diameter x (width + height)
EPS32 = 2.0 ** -24
COS_RES = (2.0 * EPS32) ** 0.5   # the angle below which float32 cosines cannot tell two directions apart
ALIGN_TOL = 2e-6    # |sin| of the angle between the rectangle's side and the hull edge


def check(oracle, pts, where=""):
    pts = np.ascontiguousarray(pts, np.int32)
    rr = oracle.min_area_rect(pts).astype(np.float64); r = oracle.min_area_rect_points(pts).astype(np.float64)
    b = oracle.mar_brute(pts)
    cx, cy, w, h, ang = rr
    k = len(b["hull"])
    # conventions of RotatedRect::points
    a = np.deg2rad(np.float64(np.float32(ang)))
    u = np.array([np.cos(a), np.sin(a)]); v = np.array([-np.sin(a), np.cos(a)]); c = np.array([cx, cy])
    want = np.array([c - w / 2 * u + h / 2 * v, c - w / 2 * u - h / 2 * v, c + w / 2 * u - h / 2 * v, c + w / 2 * u + h / 2 * v])
    scale = max(np.abs(r).max(), w, h, 1.0)
    assert np.abs(r - want).max() <= 4e-6 * scale, (where, "corner order / formula", r, want)
    if k < 3:   # a point or a segment: zero area, nothing to align (OpenCV: height 0, width = the segment)
        assert min(w, h) == 0.0 and b["min_area"] == 0.0, (where, rr)
        if k == 2:
            L = np.linalg.norm((b["hull"][1] - b["hull"][0]).astype(np.float64))
            assert abs(max(w, h) - L) <= 1e-6 * max(L, 1), (where, rr, L)
        return 0.0, 0, "rounding"
    assert -90.0 <= ang <= 0.0, (where, "angle convention", ang)
    f64 = oracle.orc().orc_min_area_rect_f64_area; f64.restype = C.c_double
    a64 = f64(pts.ctypes.data_as(C.c_void_p), len(pts))
    assert abs(a64 - b["min_area"]) <= 1e-9 * max(b["min_area"], 1.0), (where, "the caliper walk in double precision is not the exhaustive minimum", a64, b["min_area"])
    area = w * h
    exact = b["min_area"]
    H = b["hull"].astype(np.float64); E = np.roll(H, -1, 0) - H; En = E / np.linalg.norm(E, axis=1, keepdims=True)
    diam = float(np.sqrt(((H[:, None] - H[None]) ** 2).sum(-1)).max())
    unit = EPS32 * diam * (w + h)
    rel = abs(area - exact) / unit
    if rel > AREA_UNITS:   # the thin-hull clause: both sides within D x sqrt(2 eps32) of the exact rectangle's, points within that of the rectangle
        i = b["best_edge"]; e = H[(i + 1) % len(H)] - H[i]; eu = e / np.linalg.norm(e); ev = np.array([-eu[1], eu[0]])
        ex = sorted((np.ptp((H - H[i]) @ eu), np.ptp((H - H[i]) @ ev))); got = sorted((w, h))
        tol = diam * COS_RES
        assert abs(got[0] - ex[0]) <= tol and abs(got[1] - ex[1]) <= tol, (where, "area", area, exact, rel, "sides", got, ex, "tolerance", tol)
        d = pts.astype(np.float64) - c
        assert np.abs(d @ u).max() <= w / 2 + 2 * tol and np.abs(d @ v).max() <= h / 2 + 2 * tol, (where, "containment (thin hull)")   # (side error + the tilt's lever over D / 2)
        return rel, b["ties"], "cosine resolution"
    # alignment: a hull edge parallel to one of the two side directions, and that edge's rectangle is (within rounding) a minimum
    sin_u = np.abs(En[:, 0] * u[1] - En[:, 1] * u[0]); sin_v = np.abs(En[:, 0] * v[1] - En[:, 1] * v[0])
    par = np.minimum(sin_u, sin_v)
    j = int(np.argmin(par))
    assert par[j] <= ALIGN_TOL, (where, "no hull edge along a side", par[j])
    assert abs(b["edge_area"][j] - exact) <= AREA_UNITS * unit, (where, "aligned edge is not a minimum", b["edge_area"][j], exact)
    # containment
    d = pts.astype(np.float64) - c
    assert np.abs(d @ u).max() <= w / 2 + 2e-6 * scale + 1e-4 and np.abs(d @ v).max() <= h / 2 + 2e-6 * scale + 1e-4, (where, "containment")
    return rel, b["ties"], "rounding"
