"""ORACLE (test infrastructure only -- never imported by the product path): CPU restatement of the image-geometry side of
the hot path, i.e. of what the reference delegates to OpenCV.

    reference call sites:  /root/reference/src/base/transforms/utils.py:25-57  (get_affine_transform -> cv2.getAffineTransform)
                           /root/reference/src/base/transforms/utils.py:89-97  (resize_align_multi_scale -> cv2.warpAffine)
                           /root/reference/src/keypoints/results.py:158-171    (transform_coords, inverse=True)
                           /root/reference/src/keypoints/model.py:46-51,70-76  (ToTensor + Normalize of the warped image)

The arithmetic lives in a third-party dependency that is absent here: opencv-python 4.9.0.80 (poetry.lock:1601-1602).
**PARITY UNPINNED**: no cv2 is installed in the build container or on the GPU box and the reference's own tests hold no image
fixtures, so this file restates OpenCV 4.9's *published* algorithm (modules/imgproc/src/imgwarp.cpp: cv::warpAffine,
hal::warpAffine, WarpAffineInvoker, initInterTab2D, remapBilinear<FixedPtCast<int, uchar, 15>>; modules/core/src/matrix_decomp.cpp:
LUImpl; modules/imgproc/src/imgwarp.cpp: cv::getAffineTransform) and is checked only against hand-derived cases
(tests/test_oracle_cpu.py).  What it is good for: the HIP kernels and the host helpers behind the C-ABI are compared with it, so
the product and the oracle are two independent statements of that algorithm.

warpAffine (8-bit, INTER_LINEAR, BORDER_CONSTANT 0), as OpenCV computes it:
  * the 2x3 matrix is inverted in double: D = 1 / (M0*M4 - M1*M3); A11 = M4*D, A22 = M0*D, M1 *= -D, M3 *= -D,
    b1 = -A11*M2 - M1*M5, b2 = -M3*M2 - A22*M5 (in this order, every product rounded);
  * source coordinates in fixed point with AB_BITS = 10: adelta[x] = round(M0*x*1024), bdelta[x] = round(M3*x*1024),
    X0 = round((M1*y + M2)*1024) + 16, Y0 = round((M4*y + M5)*1024) + 16 (round = cvRound = nearest-even; 16 = 1024/32/2);
    X = (X0 + adelta[x]) >> 5, Y = (Y0 + bdelta[x]) >> 5: integer pixel = X >> 5 (saturated to int16), fraction = X & 31 (INTER_BITS = 5);
  * weights from the 32x32 table of int16 quadruples that sum to 32768: w = round((1-fy)(1-fx)*32768), ... ; entry (0,0) would be
    32768, which saturates to 32767, and the table builder repairs the sum by adding 1 to the bottom-right weight;
  * out = (p00*w0 + p01*w1 + p10*w2 + p11*w3 + 16384) >> 15, taps outside the image = 0.
"""
from __future__ import annotations

import numpy as np

AB_BITS, INTER_BITS = 10, 5
INTER_TAB_SIZE = 1 << INTER_BITS
COEF_BITS = 15


def cv_round(v):
    """cvRound: nearest, ties to even (lrint) -> int64"""
    return np.rint(np.asarray(v, np.float64)).astype(np.int64)


def bilinear_tab_i() -> np.ndarray:
    """initInterTab2D(INTER_LINEAR, fixpt=true): int16 [32*32, 4] = (w00, w01, w10, w11) per (fy, fx)."""
    a = np.arange(INTER_TAB_SIZE, dtype=np.float32) * np.float32(1.0 / INTER_TAB_SIZE)
    one = np.stack([np.float32(1) - a, a], 1)  # interpolateLinear: {1 - x, x} in float
    tab = np.zeros((INTER_TAB_SIZE * INTER_TAB_SIZE, 4), np.int64)
    for i in range(INTER_TAB_SIZE):
        for j in range(INTER_TAB_SIZE):
            v = (one[i][:, None] * one[j][None, :]).astype(np.float32) * np.float32(1 << COEF_BITS)
            w = np.minimum(cv_round(v), 32767).reshape(4)  # saturate_cast<short>
            diff = int(w.sum()) - (1 << COEF_BITS)
            if diff:
                # the repair loop scans itab[k1*2 + k2] for k1, k2 in {1, 2}: index 3 of this entry and (past its end) the first
                # three values of the NEXT entry, which the static table still holds as zeros when entry (0,0) -- the only one
                # that needs repair -- is built; with diff < 0 the maximum (index 3 itself: nothing is greater) takes -diff
                assert (i, j) == (0, 0) and diff == -1
                w[3] -= diff
            tab[i * INTER_TAB_SIZE + j] = w
    assert (tab.sum(1) == 1 << COEF_BITS).all()
    return tab.astype(np.int16)


_TAB = None


def invert_affine(m) -> np.ndarray:
    """cv::warpAffine without WARP_INVERSE_MAP: the double-precision inverse it builds of the forward 2x3 matrix."""
    M = np.array(m, np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M.reshape(2, 3)


def warp_affine(image: np.ndarray, m, size) -> np.ndarray:
    """cv2.warpAffine(image, m, size): uint8 HWC (or HW), m maps source -> destination, size = (w_out, h_out)."""
    global _TAB
    if _TAB is None:
        _TAB = bilinear_tab_i().astype(np.int64)
    img = image if image.ndim == 3 else image[..., None]
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    w_out, h_out = int(size[0]), int(size[1])
    M = invert_affine(m).reshape(6)
    scale = float(1 << AB_BITS)
    xs = np.arange(w_out, dtype=np.float64)
    ys = np.arange(h_out, dtype=np.float64)
    adelta = cv_round(M[0] * xs * scale)
    bdelta = cv_round(M[3] * xs * scale)
    round_delta = (1 << AB_BITS) // INTER_TAB_SIZE // 2
    X0 = cv_round((M[1] * ys + M[2]) * scale) + round_delta
    Y0 = cv_round((M[4] * ys + M[5]) * scale) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)  # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    wq = _TAB[(Y & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (X & (INTER_TAB_SIZE - 1))]  # [h_out, w_out, 4]

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
        return img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64) * ok[..., None]

    acc = (tap(sy, sx) * wq[..., 0:1] + tap(sy, sx + 1) * wq[..., 1:2] + tap(sy + 1, sx) * wq[..., 2:3] + tap(sy + 1, sx + 1) * wq[..., 3:4])
    out = np.clip((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS, 0, 255).astype(np.uint8)
    return out if image.ndim == 3 else out[..., 0]


def lu_solve(A: np.ndarray, b: np.ndarray) -> np.ndarray:
    """cv::solve(A, b, DECOMP_LU) for a small system = hal::LU64f -> LUImpl (partial pivoting, eps = DBL_EPSILON*100), in place."""
    A = np.array(A, np.float64)
    b = np.array(b, np.float64).reshape(-1)
    m = A.shape[0]
    eps = np.finfo(np.float64).eps * 100
    for i in range(m):
        k = i
        for j in range(i + 1, m):
            if abs(A[j, i]) > abs(A[k, i]):
                k = j
        if abs(A[k, i]) < eps:
            raise np.linalg.LinAlgError("singular")
        if k != i:
            A[[i, k], i:] = A[[k, i], i:]
            b[[i, k]] = b[[k, i]]
        d = -1.0 / A[i, i]
        for j in range(i + 1, m):
            alpha = A[j, i] * d
            for c in range(i + 1, m):
                A[j, c] += alpha * A[i, c]
            b[j] += alpha * b[i]
    for i in range(m - 1, -1, -1):
        s = b[i]
        for k in range(i + 1, m):
            s -= A[i, k] * b[k]
        b[i] = s / A[i, i]
    return b


def cv_get_affine_transform(src, dst) -> np.ndarray:
    """cv::getAffineTransform(Point2f src[3], Point2f dst[3]) -> 2x3 float64: the 6x6 system of imgwarp.cpp solved by LU."""
    src = np.asarray(src, np.float32)
    dst = np.asarray(dst, np.float32)
    a = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    for i in range(3):
        a[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        a[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return lu_solve(a, b).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, shift=(0, 0), inverse=False) -> np.ndarray:
    """utils.py:25-57 with the numpy dtypes the reference uses (float32 point arrays filled from float64 expressions)."""
    shift = np.array(shift)
    scale = np.array(scale)
    center = np.array(center)
    src_w = scale[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    p = [0, -src_w / 2]
    src_dir = (p[0] * cs - p[1] * sn, p[0] * sn + p[1] * cs)
    dst_dir = np.array([0, -dst_w / 2], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], np.float32)

    src[2:, :] = third(src[0, :], src[1, :])
    dst[2:, :] = third(dst[0, :], dst[1, :])
    if inverse:
        src, dst = dst, src
    return cv_get_affine_transform(src, dst)


def get_multi_scale_size(image_hw, input_size, current_scale, min_scale):
    """utils.py:60-86 on (h, w)"""
    h, w = image_hw
    center = (int(w / 2.0 + 0.5), int(h / 2.0 + 0.5))
    min_input_size = int((min_scale * input_size + 63) // 64 * 64)
    if w < h:
        w_resized = int(min_input_size * current_scale / min_scale)
        h_resized = int(int((min_input_size / w * h + 63) // 64 * 64) * current_scale / min_scale)
        scale_w = w
        scale_h = h_resized / w_resized * w
    else:
        h_resized = int(min_input_size * current_scale / min_scale)
        w_resized = int(int((min_input_size / h * w + 63) // 64 * 64) * current_scale / min_scale)
        scale_h = h
        scale_w = w_resized / h_resized * h
    return (w_resized, h_resized), center, (scale_w, scale_h)


def resize_align_multi_scale(image, input_size, current_scale, min_scale):
    """utils.py:89-97"""
    size, center, scale = get_multi_scale_size(image.shape[:2], input_size, current_scale, min_scale)
    trans = get_affine_transform(center, scale, 0, size)
    return warp_affine(image, trans, size), center, scale


def prepare_input(image, input_size, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """keypoints/model.py:70-76: resize-align, ToTensor (uint8 -> float32 / 255, CHW), Normalize -> float32 [3, h, w]"""
    resized, center, scale = resize_align_multi_scale(image, input_size, 1, 1)
    x = resized.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    x = (x - np.asarray(mean, np.float32)[:, None, None]) / np.asarray(std, np.float32)[:, None, None]
    return x.astype(np.float32), resized, center, scale


def transform_coords(kpts_coords, center, scale, output_size) -> np.ndarray:
    """results.py:158-171: float32 [K, 2+] rows, the first two columns replaced by M_inverse @ (x, y, 1) (float64 product
    stored into the float32 copy)."""
    out = np.array(kpts_coords, np.float32, copy=True)
    M = get_affine_transform(center, scale, 0, output_size, inverse=True)
    for i in range(out.shape[0]):
        x, y = out[i, :2].tolist()
        out[i, :2] = np.dot(M, np.array([x, y, 1.0]).T)[:2]
    return out
