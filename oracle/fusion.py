"""Oracle: depth-map fusion filter.

Test infrastructure (see oracle/__init__.py): numpy restatement of eval_rcmvsnet_dtu.py:281-338 (reproject_with_depth,
check_geometric_consistency) and of the per-reference-view body of filter_depth (:369-425), with numpy's dtype promotion
kept as the reference has it (float32 camera matrices inverted / multiplied in float32, the per-pixel chain in float64, the
float32 casts at the same places).  Pinned by tests/golden/fusion.npz, which tests/golden/make_golden.py --only-fusion
produced by importing the reference's eval script.

One step cannot be pinned by running the reference: it samples the source depth with ``cv2.remap(..., INTER_LINEAR)``, and
opencv-python (4.5.5.62, requirements.txt:31) is a third-party dependency absent from the reference tree and from this image.
``remap_linear`` below restates its published algorithm (modules/imgproc/src/imgwarp.cpp, cv::remap -> remapBilinear with the
initInterTab2D tables: coordinates rounded to 1/32 pixel with cvRound = round-half-even, a 32 x 32 float32 weight table,
BORDER_CONSTANT value 0), the golden generator hands the same function to the reference as ``cv2.remap``, and the restatement
is pinned by KNOWN ANSWERS worked out from that algorithm in exact arithmetic (integer / 1-32nd-grid coordinates, ties,
half-outside footprints, non-finite coordinates): tests/test_cv_known_answers_cpu.py.  Everything around that call is pinned
by the reference's own code.
"""
import numpy as np


def remap_linear(img, map_x, map_y):
    """cv2.remap(img, map_x, map_y, interpolation=cv2.INTER_LINEAR) for float32 single-channel input, default border."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape
    fx = np.asarray(map_x, np.float32) * np.float32(32.0)
    fy = np.asarray(map_y, np.float32) * np.float32(32.0)
    far = ~(np.abs(fx) < 1.0e9) | ~(np.abs(fy) < 1.0e9)
    sx = np.rint(np.where(far, 0, fx)).astype(np.int64)
    sy = np.rint(np.where(far, 0, fy)).astype(np.int64)
    ix, iy = sx >> 5, sy >> 5
    ax = (sx & 31).astype(np.float32) * np.float32(1.0 / 32.0)
    ay = (sy & 31).astype(np.float32) * np.float32(1.0 / 32.0)
    one = np.float32(1.0)
    w = [(one - ay) * (one - ax), (one - ay) * ax, ay * (one - ax), ay * ax]
    out = np.zeros(fx.shape, np.float32)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        xx, yy = ix + dx, iy + dy
        inside = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        s = np.where(inside, img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0.0)).astype(np.float32)
        out = out + s * w[k] if k else s * w[k]
    out[far] = 0.0
    return out.astype(np.float32)


def reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src):
    """eval_rcmvsnet_dtu.py:281-321."""
    h, w = depth_ref.shape
    xr, yr = np.meshgrid(np.arange(0, w), np.arange(0, h))
    xr, yr = xr.reshape(-1), yr.reshape(-1)
    ones = np.ones_like(xr)
    p_ref = np.linalg.inv(K_ref) @ (np.vstack((xr, yr, ones)) * depth_ref.reshape(-1))
    p_src = ((E_src @ np.linalg.inv(E_ref)) @ np.vstack((p_ref, ones)))[:3]
    k = K_src @ p_src
    xy_src = k[:2] / k[2:3]
    x_src = xy_src[0].reshape(h, w).astype(np.float32)
    y_src = xy_src[1].reshape(h, w).astype(np.float32)
    sampled = remap_linear(depth_src, x_src, y_src)
    p_src = np.linalg.inv(K_src) @ (np.vstack((xy_src, ones)) * sampled.reshape(-1))
    p_back = ((E_ref @ np.linalg.inv(E_src)) @ np.vstack((p_src, ones)))[:3]
    depth_back = p_back[2].reshape(h, w).astype(np.float32)
    k = K_ref @ p_back
    xy_back = k[:2] / k[2:3]
    return depth_back, xy_back[0].reshape(h, w).astype(np.float32), xy_back[1].reshape(h, w).astype(np.float32), x_src, y_src


def check_geometric_consistency(depth_ref, K_ref, E_ref, depth_src, K_src, E_src, img_dist_thresh, depth_thresh):
    """eval_rcmvsnet_dtu.py:324-338."""
    h, w = depth_ref.shape
    xr, yr = np.meshgrid(np.arange(0, w), np.arange(0, h))
    depth_back, x_back, y_back, x_src, y_src = reproject_with_depth(depth_ref, K_ref, E_ref, depth_src, K_src, E_src)
    with np.errstate(divide="ignore", invalid="ignore"):
        dist = np.sqrt((x_back - xr) ** 2 + (y_back - yr) ** 2)
        rel = np.abs(depth_back - depth_ref) / depth_ref
        mask = np.logical_and(dist < img_dist_thresh, rel < depth_thresh)
    depth_back[~mask] = 0
    return mask, depth_back, x_src, y_src


def fuse_view(depth_ref, conf, img, K_ref, E_ref, src_depths, src_K, src_E, prob_threshold, num_consistent, img_dist_thresh, depth_thresh):
    """Per-reference-view body of filter_depth (eval_rcmvsnet_dtu.py:369-425).  Returns a dict with the three masks, the
    averaged depth (float64, as numpy promotes it), and the surviving world points (float32) and colours (uint8)."""
    photo = conf > prob_threshold
    geo_sum = 0
    reprojected = []
    for d, K, E in zip(src_depths, src_K, src_E):
        m, back, _, _ = check_geometric_consistency(depth_ref, K_ref, E_ref, d, K, E, img_dist_thresh, depth_thresh)
        geo_sum = geo_sum + m.astype(np.int32)
        reprojected.append(back)
    avg = (sum(reprojected) + depth_ref) / (geo_sum + 1)
    geo = geo_sum >= num_consistent
    final = np.logical_and(photo, geo)
    h, w = avg.shape
    x, y = np.meshgrid(np.arange(0, w), np.arange(0, h))
    x, y, d = x[final], y[final], avg[final]
    p = np.linalg.inv(K_ref) @ (np.vstack((x, y, np.ones_like(x))) * d)
    world = (np.linalg.inv(E_ref) @ np.vstack((p, np.ones_like(x))))[:3]
    return {"photo": photo, "geo": geo, "final": final, "depth_avg": avg, "geo_sum": geo_sum,
            "xyz": world.transpose(1, 0).astype(np.float32), "rgb": (img[final] * 255).astype(np.uint8)}


def ply_bytes(xyz, rgb):
    """The bytes plyfile 0.7.4 (requirements.txt:34; absent here) writes for PlyData([PlyElement.describe(vertex_all, 'vertex')])
    with properties x y z (f4) and red green blue (u1): binary little-endian, no comments (eval_rcmvsnet_dtu.py:433-446)."""
    n = len(xyz)
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
            "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % n).encode("ascii")
    rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["red"], rec["green"], rec["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    return head + rec.tobytes()
