"""Shared helpers of the parity tests: run the HIP library and the CPU oracle side by side."""
import numpy as np

from oracle import oracle as orc_mod
from semantic_dsp_map_amd import binding

STATE_KEYS = [k for k, _ in binding.STATE_FIELDS]
ST_INVALID = 0


def snapshot(m):
    return {"state": m.dump_state(), "ring": m.ring_state(), "stamps": m.stamps()}


def restore(m, snap):
    m.load_state(snap["state"])
    m.set_stamps(*snap["stamps"])
    m.set_ring_state(snap["ring"])


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a


def diff_report(name, a, b, extra=None, limit=8):
    a, b = np.asarray(a), np.asarray(b)
    bad = np.flatnonzero(bits(a).ravel() != bits(b).ravel())
    if bad.size == 0:
        return None
    lines = ["%s: %d of %d entries differ" % (name, bad.size, a.size)]
    for i in bad[:limit]:
        s = "  [%d] oracle=%r gpu=%r" % (i, a.ravel()[i], b.ravel()[i])
        if extra is not None:
            s += " " + extra(int(i))
        lines.append(s)
    if a.dtype == np.float32:
        with np.errstate(invalid="ignore"):
            lines.append("  max |diff| = %g" % np.nanmax(np.abs(a.ravel()[bad].astype(np.float64) - b.ravel()[bad])))
    return "\n".join(lines)


def compare_maps(o, g, S, check_results=True, check_bins=False, tag=""):
    """Returns a list of human-readable mismatch reports (empty = bit-identical)."""
    out = []
    so, sg = o.dump_state(), g.dump_state()

    def slot_info(i):
        return "(voxel %d slot %d | o: st=%d ts=%d trk=%d w=%g | g: st=%d ts=%d trk=%d w=%g)" % (
            i // S, i % S, so["status"][i], so["ts"][i], so["track"][i], so["w"][i],
            sg["status"][i], sg["ts"][i], sg["track"][i], sg["w"][i])

    for k in STATE_KEYS:
        r = diff_report(tag + "state." + k, so[k], sg[k], slot_info)
        if r:
            out.append(r)
    ro, rg = o.ring_state(), g.ring_state()
    for k in ro:
        if k == "last_pos" or k == "map_center":
            if not np.array_equal(bits(np.float32(ro[k])), bits(np.float32(rg[k]))):
                out.append("%sring.%s: oracle=%r gpu=%r" % (tag, k, ro[k], rg[k]))
        elif ro[k] != rg[k]:
            out.append("%sring.%s: oracle=%r gpu=%r" % (tag, k, ro[k], rg[k]))
    for name, a, b in zip("xyz", o.stamps(), g.stamps()):
        r = diff_report(tag + "stamps_" + name, a, b)
        if r:
            out.append(r)
    if check_bins:
        r = diff_report(tag + "bin_counts", o.bin_counts(), g.bin_counts())
        if r:
            out.append(r)
        else:
            r = diff_report(tag + "bins", o.bins(), g.bins())
            if r:
                out.append(r)
    if check_results:
        vo, vg = o.voxels(), g.voxels()
        for k in ("occ", "label", "track", "wsum"):
            r = diff_report(tag + "voxels." + k, vo[k], vg[k])
            if r:
                out.append(r)
    return out


def make_pair(cfg, params, noise, bin_order=1, **gpu_kw):
    o = orc_mod.OracleMap(dict(cfg, bin_order=bin_order), params, noise)
    g = binding.SdmMap(cfg, params, noise, **gpu_kw)
    return o, g
