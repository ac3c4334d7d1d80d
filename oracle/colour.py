"""CPU restatement of the colouring of the emitted cloud - test infrastructure only.

Reference: SemanticDSPMap::getOccupancyResult, include/semantic_dsp_map.h:1274-1376 (colour rules), :45-63 (the two
colour maps), utils/data_base.h:216-232 (label colours).  The last step there is an 8-bit RGB -> HSV -> RGB round trip
through OpenCV's cvtColor (semantic_dsp_map.h:1333-1351), which is NOT an identity and dims V by 0.7 for voxels
outside the view.  OpenCV is a dependency the reference does not vendor and does not pin (CMakeLists.txt:
find_package(OpenCV REQUIRED)); what follows restates the published 8-bit algorithms of OpenCV 4.x,
modules/imgproc/src/color_hsv.simd.hpp:
  RGB2HSV_b   integer arithmetic, hsv_shift = 12, tables sdiv_table[i] = cvRound((255 << 12) / i),
              hdiv_table180[i] = cvRound((180 << 12) / (6 i)), h range 180
  HSV2RGB_b   per pixel in float: h * (6/180), s / 255, v / 255 -> HSV2RGB_native (sector table) -> x 255,
              saturate_cast<uchar> (round half to even)
Parity unpinned: no OpenCV in this image to check the restatement against; tests/test_colour.py holds hand-computed
known answers for it and the HIP kernel is compared with it bit for bit."""
import numpy as np

HSV_SHIFT = 12
_i = np.arange(1, 256, dtype=np.float64)
SDIV = np.zeros(256, np.int64)
HDIV180 = np.zeros(256, np.int64)
SDIV[1:] = np.rint((255 << HSV_SHIFT) / (1.0 * _i)).astype(np.int64)
HDIV180[1:] = np.rint((180 << HSV_SHIFT) / (6.0 * _i)).astype(np.int64)


def jet_256():
    """color_map_jet_256_, semantic_dsp_map.h:50-63 (r, g, b)."""
    out = np.zeros((256, 3), np.int64)
    for i in range(256):
        if i < 64:
            out[i] = (0, 0, i * 4)
        elif i < 128:
            out[i] = (0, (i - 64) * 4, 255)
        elif i < 192:
            out[i] = ((i - 128) * 4, 255, 255 - (i - 128) * 4)
        else:
            out[i] = (255, 255 - (i - 192) * 4, 0)
    return out


def rgb2hsv_8u(rgb):
    """RGB2HSV_b (color_hsv.simd.hpp), scalar path, hrange 180.  rgb: (n, 3) uint8 -> (n, 3) uint8 (h, s, v)."""
    rgb = np.asarray(rgb, np.int64)
    r, g, b = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    v = np.maximum(np.maximum(b, g), r)
    vmin = np.minimum(np.minimum(b, g), r)
    diff = v - vmin
    vr = v == r
    vg = v == g
    s = (diff * SDIV[v] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT
    h = np.where(vr, g - b, np.where(vg, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * HDIV180[diff] + (1 << (HSV_SHIFT - 1))) >> HSV_SHIFT       # arithmetic shift: floor, also for h < 0
    h = h + np.where(h < 0, 180, 0)
    return np.stack([np.clip(h, 0, 255), s, v], 1).astype(np.uint8)


_SECTOR = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])  # (b, g, r) <- tab[]


def hsv2rgb_8u(hsv):
    """HSV2RGB_b -> HSV2RGB_native in float32 (color_hsv.simd.hpp).  hsv: (n, 3) uint8 -> (n, 3) uint8 (r, g, b)."""
    hsv = np.asarray(hsv)
    f = np.float32
    h = hsv[:, 0].astype(f)
    s = hsv[:, 1].astype(f) * f(1.0 / 255.0)
    v = hsv[:, 2].astype(f) * f(1.0 / 255.0)
    hh = np.fmod(h * f(6.0 / 180.0), f(6.0)).astype(f)
    sector = np.floor(hh).astype(np.int64)
    hh = (hh - sector.astype(f)).astype(f)
    bad = (sector < 0) | (sector >= 6)
    sector = np.where(bad, 0, sector)
    hh = np.where(bad, f(0), hh).astype(f)
    one = f(1.0)
    tab = np.stack([v, (v * (one - s)).astype(f), (v * (one - (s * hh).astype(f))).astype(f),
                    (v * (one - (s * (one - hh).astype(f)).astype(f))).astype(f)], 1)
    idx = _SECTOR[sector]
    n = np.arange(len(v))
    bgr = np.stack([tab[n, idx[:, 0]], tab[n, idx[:, 1]], tab[n, idx[:, 2]]], 1)
    grey = s == 0
    bgr = np.where(grey[:, None], v[:, None], bgr)
    out = np.clip(np.rint((bgr * f(255.0)).astype(f)), 0, 255).astype(np.uint8)   # saturate_cast<uchar>: round half to even
    return out[:, ::-1].copy()


def colour_points(z, y, track, label, occ, out_of_fov, label_bgr, perm, background_label, max_movable, colour_by_label=False,
                  jet_axis=0, evaluation_format=False):
    """Colour rules of semantic_dsp_map.h:1274-1351 for occupied voxels (occ 1 = occupied, 2 = guessed occupied).
    z, y: emitted coordinates (after the optional camera-centre subtraction).  Returns (n, 3) uint8 (r, g, b)."""
    n = len(z)
    rgb = np.zeros((n, 3), np.int64)
    jet = jet_256()
    f = np.float32
    src = (-np.asarray(z, f) + f(2.0)) if jet_axis == 0 else (np.asarray(y, f) + f(2.0))
    ci = np.clip(((src * f(51.2)).astype(f)).astype(np.int64), 0, 255)             # static_cast<int>: truncation
    track = np.asarray(track, np.int64)
    label = np.asarray(label, np.int64)
    occ = np.asarray(occ)
    is_bg = label == background_label
    static = (track > max_movable) | bool(colour_by_label)
    for k in range(n):
        if occ[k] != 1:                       # guessed occupied: white (:1325-1331)
            rgb[k] = (255, 255, 255)
        elif is_bg[k]:                        # :1277-1294
            rgb[k] = (0, 0, 0) if evaluation_format else jet[ci[k]]
        elif static[k]:                       # :1297-1309 (BGR -> RGB)
            c = label_bgr[label[k]]
            rgb[k] = (c[2], c[1], c[0])
        elif evaluation_format:               # :1311-1316
            rgb[k] = (label[k], track[k] >> 8, track[k] & 0xFF)
        else:                                 # :1317-1319; PINNED: the 256-entry table is indexed with track & 255
            rgb[k] = (160, perm[track[k] & 0xFF], perm[label[k]])
    rgb = rgb.astype(np.uint8)
    if evaluation_format:
        return rgb
    hsv = rgb2hsv_8u(rgb)
    dim = np.asarray(out_of_fov, bool)
    hsv[dim, 2] = (hsv[dim, 2].astype(np.float32) * np.float32(0.7)).astype(np.uint8)   # uchar *= 0.7f: truncation (:1342)
    return hsv2rgb_8u(hsv)
