"""Seeded synthetic workloads: RGB-D frames and template banks.

Host-side data generation only (numpy); nothing here is on the measured path.  The same generators
feed the parity tests (HIP vs oracle on identical inputs) and bench.py.
"""
import numpy as np

FEATURE_DTYPE = np.dtype([("x", "<i4"), ("y", "<i4"), ("label", "<i4")])
DESC_DTYPE = np.dtype([("width", "<i4"), ("height", "<i4"), ("pyramid_level", "<i4"), ("num_features", "<i4")])


def make_frame(width=640, height=480, seed=1234, n_shapes=40, noise_sigma=2.0, holes=0.05):
    """Colour: random filled ellipses / rectangles / triangles with uniform BGR fill + Gaussian noise
    (edges that survive the >=5/9 orientation vote).  Depth: piecewise planes 500-1200 mm following
    the same shapes + `holes` fraction of zero pixels.  Returns (bgr u8 [h,w,3], depth u16 [h,w])."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    bgr = np.empty((height, width, 3), np.float32)
    bgr[:] = rng.integers(40, 216, 3)
    # float64 from here on: the planes below mix in float64 scalars
    depth = (900.0 + 0.15 * (xx - width / 2) + 0.1 * (yy - height / 2)).astype(np.float64)
    for _ in range(n_shapes):
        kind = rng.integers(0, 3)
        cx, cy = rng.uniform(0, width), rng.uniform(0, height)
        sx, sy = rng.uniform(0.04, 0.22) * width, rng.uniform(0.04, 0.22) * height
        # every shape lies inside this box, so only the box is evaluated (same pixels as the full image would give)
        rad = float(np.hypot(sx, sy)) + 2.0
        x0, x1 = max(int(cx - rad), 0), min(int(cx + rad) + 2, width)
        y0, y1 = max(int(cy - rad), 0), min(int(cy + rad) + 2, height)
        X, Y = xx[y0:y1, x0:x1], yy[y0:y1, x0:x1]
        if kind == 0:
            m = ((X - cx) / sx) ** 2 + ((Y - cy) / sy) ** 2 <= 1.0
        elif kind == 1:
            a = rng.uniform(0, np.pi)
            u = (X - cx) * np.cos(a) + (Y - cy) * np.sin(a)
            v = -(X - cx) * np.sin(a) + (Y - cy) * np.cos(a)
            m = (np.abs(u) <= sx) & (np.abs(v) <= sy)
        else:
            p = np.stack([rng.uniform(cx - sx, cx + sx, 3), rng.uniform(cy - sy, cy + sy, 3)], 1)
            def side(a, b):
                return (X - a[0]) * (b[1] - a[1]) - (Y - a[1]) * (b[0] - a[0])
            s0, s1, s2 = side(p[0], p[1]), side(p[1], p[2]), side(p[2], p[0])
            m = ((s0 >= 0) & (s1 >= 0) & (s2 >= 0)) | ((s0 <= 0) & (s1 <= 0) & (s2 <= 0))
        bgr[y0:y1, x0:x1][m] = rng.integers(0, 256, 3)
        gx, gy = rng.uniform(-0.6, 0.6, 2)
        plane = rng.uniform(500, 1200) + gx * (X - cx) + gy * (Y - cy)
        depth[y0:y1, x0:x1] = np.where(m, plane, depth[y0:y1, x0:x1])
    bgr += rng.normal(0.0, noise_sigma, bgr.shape).astype(np.float32)
    bgr = np.clip(np.rint(bgr), 0, 255).astype(np.uint8)
    depth = np.clip(np.rint(depth), 300, 3000).astype(np.uint16)
    depth[rng.random((height, width)) < holes] = 0
    return bgr, depth


def _ring_points(rng, n, w, h):
    """n points on a jittered closed contour inside [0,w]x[0,h] (colour features sit on object outlines)."""
    t = np.sort(rng.uniform(0, 2 * np.pi, n))
    r = 0.5 * (0.75 + 0.25 * np.sin(3 * t + rng.uniform(0, 6.28)))
    x = np.clip(np.rint(w / 2 + r * w * np.cos(t)), 0, w).astype(np.int32)
    y = np.clip(np.rint(h / 2 + r * h * np.sin(t)), 0, h).astype(np.int32)
    return x, y


def make_bank(n_templates, num_modalities=2, pyramid_levels=2, seed=4321, num_features=63, fixed_l0_size=None,
              size_range=(48, 160), quantized=None, crop_fraction=0.1, frame_size=(640, 480), T0=5):
    """Template bank as (descs, features) in the lm_add_class layout.

    fixed_l0_size=(w, h): every template has that level-0 bbox (the fixed-geometry roofline variant of
    SURVEY.md 8d: 96x96 at level 0 = 48x48 at level 1).  Otherwise w, h ~ U{size_range}, even.
    quantized: optional {(level, modality): u8 image}; when given, `crop_fraction` of the templates
    are cut out of those quantised images (features = pixels whose one-hot label is set), so real
    matches with similarity 100 exist.  Returns (descs, features, crops) where crops lists
    (template_id, x0, y0) of the crop templates."""
    rng = np.random.default_rng(seed)
    M, L = num_modalities, pyramid_levels
    descs = np.zeros(n_templates * L * M, DESC_DTYPE)
    feats = []
    crops = []
    fw, fh = frame_size
    for t in range(n_templates):
        if fixed_l0_size is not None:
            w0, h0 = fixed_l0_size
        else:
            w0 = int(rng.integers(size_range[0] // 2, size_range[1] // 2 + 1)) * 2
            h0 = int(rng.integers(size_range[0] // 2, size_range[1] // 2 + 1)) * 2
        is_crop = quantized is not None and rng.random() < crop_fraction
        x0 = y0 = 0
        if is_crop:
            border = 8 * T0 + 8
            x0 = int(rng.integers(border // 2, (fw - w0 - border) // 2)) * 2
            y0 = int(rng.integers(border // 2, (fh - h0 - border) // 2)) * 2
        tmpl_feats = []
        ok = True
        for l in range(L):
            wl, hl = w0 >> l, h0 >> l
            nf = num_features >> l
            for m in range(M):
                if is_crop:
                    q = quantized[(l, m)]
                    sub = q[(y0 >> l):(y0 >> l) + hl + 1, (x0 >> l):(x0 >> l) + wl + 1]
                    ys, xs = np.nonzero(sub)
                    if ys.size < nf:
                        ok = False
                        break
                    pick = rng.choice(ys.size, nf, replace=False)
                    f = np.zeros(nf, FEATURE_DTYPE)
                    f["x"], f["y"] = xs[pick], ys[pick]
                    f["label"] = np.log2(sub[ys[pick], xs[pick]].astype(np.float64)).astype(np.int32)
                else:
                    f = np.zeros(nf, FEATURE_DTYPE)
                    if m == 0:
                        f["x"], f["y"] = _ring_points(rng, nf, wl, hl)
                    else:
                        f["x"] = rng.integers(wl // 6, wl - wl // 6 + 1, nf)
                        f["y"] = rng.integers(hl // 6, hl - hl // 6 + 1, nf)
                    f["label"] = rng.integers(0, 8, nf)
                tmpl_feats.append(f)
            if not ok:
                break
        if not ok:  # not enough structure in the crop: fall back to a random template
            is_crop = False
            tmpl_feats = []
            for l in range(L):
                wl, hl = w0 >> l, h0 >> l
                nf = num_features >> l
                for m in range(M):
                    f = np.zeros(nf, FEATURE_DTYPE)
                    f["x"] = rng.integers(0, wl + 1, nf)
                    f["y"] = rng.integers(0, hl + 1, nf)
                    f["label"] = rng.integers(0, 8, nf)
                    tmpl_feats.append(f)
        for l in range(L):
            for m in range(M):
                k = (t * L + l) * M + m
                descs[k] = (w0 >> l, h0 >> l, l, tmpl_feats[l * M + m].size)
        feats.extend(tmpl_feats)
        if is_crop:
            crops.append((t, x0, y0))
    features = np.concatenate(feats) if feats else np.zeros(0, FEATURE_DTYPE)
    return descs, features, crops


def algorithmic_scan_bytes(descs, features, num_modalities, pyramid_levels, frame_size, T_low):
    """B_sim = sum_t sum_m F_{m,low}(t) * P(t), with P = span_y*W + span_x + 1."""
    M, L = num_modalities, pyramid_levels
    w = frame_size[0] >> (L - 1)
    h = frame_size[1] >> (L - 1)
    W, H = w // T_low, h // T_low
    d = descs.reshape(-1, L, M)
    low = d[:, L - 1, :]
    wf = (low["width"][:, 0] - 1) // T_low + 1
    hf = (low["height"][:, 0] - 1) // T_low + 1
    P = np.clip((H - hf) * W + (W - wf) + 1, 0, W * H)
    return float((low["num_features"].sum(axis=1) * P).sum())
