"""Known-answer tests derived from the reference code, written against the common Python
surface of the oracle (oracle.OracleMap) and the HIP library (binding.SdmMap): load_state / update(stop_after)
/ dump_state / voxels / ring_state / stamps.  tests/test_oracle_kat.py runs them on the CPU oracle,
tests/test_kat_gpu.py on the GPU through the C ABI.

Each case builds a tiny hand-made map state, runs one frame (or a prefix of its stages) and compares with
numbers worked out by hand from the reference's formulas (cited per case).
"""
import numpy as np

F = np.float32

# 16^3 voxels of 0.5 m, 8 slots; tiny pinhole camera looking along +z from the origin.
K0 = dict(x_n=4, y_n=4, z_n=4, p_n=3, voxel_size=0.5, fx=20.0, fy=20.0, cx=16.0, cy=12.0, width=32, height=24,
          depth_min=0.3, depth_max=6.0, window_half=1, max_movable_track=65522)
PARAMS = dict(detection_probability=0.9, noise_number=0.05, nb_ptc_num_per_point=1, occupancy_threshold=0.3,
              max_obersevation_lost_time=5, forgetting_rate=1.0, max_forget_count=5, match_score_threshold=0.3,
              id_transition_probability=0.1, if_consider_depth_noise=1, if_use_independent_filter=0,
              depth_noise_first_order=0.0, depth_noise_zero_order=0.2)
LP = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("sigma", "<f4"), ("track_id", "<u2"), ("label_id", "u1"),
               ("is_valid", "u1")])
STATE = [("px", np.float32), ("py", np.float32), ("pz", np.float32), ("w", np.float32), ("ts", np.uint16),
         ("track", np.uint16), ("label", np.uint8), ("status", np.uint8), ("forget", np.uint8), ("owner", np.uint16)]
INVALID, UPDATED, REGULAR_BORN, GUESSED_BORN, COPIED, TIMEPTC = range(6)
IDENT_Q = np.array([1, 0, 0, 0], np.float32)
ORIGIN = np.zeros(3, np.float32)


def empty_state(cfg):
    V, S = 1 << (cfg["x_n"] + cfg["y_n"] + cfg["z_n"]), 1 << cfg["p_n"]
    st = {k: np.zeros(V * S, dt) for k, dt in STATE}
    st["owner"][:] = 0xFFFF
    st["status"].reshape(V, S)[:, 0] = TIMEPTC
    return st


def ring0(gts=1):
    return {"global_time_stamp": gts, "moved_steps": [0, 0, 0], "eq_steps": [0, 0, 0], "map_center": [0.0, 0.0, 0.0],
            "last_pos": [0.0, 0.0, 0.0], "birth_cursor": 0, "move_cursor": 0}


def voxel_of(cfg, x, y, z):
    """storage voxel index of a position with no ring shift (operations.h:864-900): floor((p - min)/size), row-major"""
    n = [1 << cfg["x_n"], 1 << cfg["y_n"], 1 << cfg["z_n"]]
    size = F(cfg["voxel_size"])
    recip = F(1.0) / size
    idx = []
    for p, N in zip((x, y, z), n):
        pmin = -F(N >> 1) * size
        f = (F(p) - pmin) * recip
        assert -1 < f < N
        idx.append(int(f))
    return ((idx[2] << cfg["y_n"]) | idx[1]) << cfg["x_n"] | idx[0]


def put(st, cfg, voxel, slot, pos, w, status=UPDATED, ts=1, track=65535, label=0, forget=0, owner=0xFFFF):
    S = 1 << cfg["p_n"]
    i = voxel * S + slot
    st["px"][i], st["py"][i], st["pz"][i] = pos
    st["w"][i], st["status"][i], st["ts"][i] = w, status, ts
    st["track"][i], st["label"][i], st["forget"][i], st["owner"][i] = track, label, forget, owner
    return i


def blank_frame(cfg, depth_value=100.0):
    n = cfg["width"] * cfg["height"]
    return np.full(n, depth_value, np.float32), np.zeros(n, LP)


def set_point(cloud, cfg, row, col, pos, sigma, track=65535, label=0):
    p = cloud[row * cfg["width"] + col:row * cfg["width"] + col + 1]
    p["x"], p["y"], p["z"] = pos
    p["sigma"], p["track_id"], p["label_id"], p["is_valid"] = sigma, track, label, 1


def fresh(make, cfg=K0, params=PARAMS, st=None, gts=1):
    m = make(cfg, params)
    m.load_state(st if st is not None else empty_state(cfg))
    m.set_ring_state(ring0(gts))
    return m


# ---------------------------------------------------------------------------------------------- (8) A4
def case_first_vacant_slot_and_full_voxel(make):
    """addParticleByGlobalPos (operations.h:782-803): births fill slots 1,2,... in raster order; a full voxel
    rejects; a point outside the map is dropped.  nb = 1 in the noise flavour: no RNG, no retry without resample."""
    cfg = K0
    m = fresh(make)
    depth, cloud = blank_frame(cfg)
    target = (0.6, 0.1, 2.1)            # all the same voxel
    v = voxel_of(cfg, *target)
    # birth raster order (semantic_dsp_map.h:778-800): pass (0,0) visits (0,0),(0,3),(0,6)...; then pass (0,1) ...
    pix = [(0, 0), (0, 3), (0, 1), (1, 0), (0, 6), (3, 3), (0, 2), (0, 4), (2, 2)]
    order = sorted(range(len(pix)), key=lambda k: ((pix[k][0] % 3) * 3 + pix[k][1] % 3, pix[k][0] // 3, pix[k][1] // 3))
    for k, (r, c) in enumerate(pix):
        set_point(cloud, cfg, r, c, (target[0] + 0.01 * k, target[1], target[2]), 0.2, label=k + 1)
    set_point(cloud, cfg, 5, 5, (100.0, 0.0, 0.0), 0.2, label=77)     # outside the map
    m.update(depth, cloud, ORIGIN, IDENT_Q, stop_after="birth")
    st = m.dump_state()
    S = 8
    labels = st["label"][v * S:(v + 1) * S]
    status = st["status"][v * S:(v + 1) * S]
    assert status[0] == TIMEPTC and np.all(status[1:] == REGULAR_BORN)
    # 9 births into a voxel with 7 free slots: the first 7 in raster order get slots 1..7
    assert list(labels[1:]) == [order[k] + 1 for k in range(7)]
    assert np.all(st["w"][v * S + 1:(v + 1) * S] == F(0.05)) and np.all(st["ts"][v * S + 1:(v + 1) * S] == 2)
    assert int((st["status"] == REGULAR_BORN).sum()) == 7 and not np.any(st["label"] == 77)


# ---------------------------------------------------------------------------------------------- (5) A9
def _resample_expect(weights, S=8):
    """resampleParticlesInVoxel (semantic_dsp_map.h:1448-1519) in float32 by hand"""
    ws = F(0)
    for w in weights:
        ws = F(ws + F(w))
    trig = S >> 1
    if len(weights) <= trig:
        return None
    if ws < F(0.01):
        return [None] * len(weights)
    wpp = F(ws / F(trig))
    if wpp > 1:
        wpp = F(1)
    run, thr, out = F(0), wpp, []
    for w in weights:
        run = F(run + F(w))
        if run < thr:
            out.append(None)
        else:
            out.append(wpp)
            thr = F(thr + wpp)
            while run > thr:
                thr = F(thr + wpp)
    return out


def case_resample(make):
    """A full voxel with 5 UPDATED + 2 REGULAR_BORN particles; one birth fails, the voxel resamples once and the
    birth is retried (semantic_dsp_map.h:1205-1227).  Second voxel: weight sum < 0.01 -> all UPDATED wiped.
    Third voxel: 4 UPDATED (= S/2, not more) -> no resample, the birth is lost."""
    cfg = K0
    S = 8
    st = empty_state(cfg)
    posA, posB, posC = (0.6, 0.1, 2.1), (-1.1, 0.1, 2.1), (1.6, 0.1, 2.1)
    vA, vB, vC = (voxel_of(cfg, *p) for p in (posA, posB, posC))
    wA = [0.30, 0.02, 0.45, 0.05, 0.08]
    for s, w in enumerate(wA, start=1):
        put(st, cfg, vA, s, posA, w, UPDATED, ts=1, track=3, owner=3)
    put(st, cfg, vA, 6, posA, 0.05, REGULAR_BORN, ts=1)
    put(st, cfg, vA, 7, posA, 0.05, REGULAR_BORN, ts=1)
    for s in range(1, 8):
        put(st, cfg, vB, s, posB, 0.001, UPDATED, ts=1)
    for s in range(1, 5):
        put(st, cfg, vC, s, posC, 0.2, UPDATED, ts=1)
    for s in range(5, 8):
        put(st, cfg, vC, s, posC, 0.05, REGULAR_BORN, ts=1)
    m = fresh(make, st=st)
    depth, cloud = blank_frame(cfg)
    set_point(cloud, cfg, 0, 0, posA, 0.2, track=3, label=9)
    set_point(cloud, cfg, 0, 3, posB, 0.2, label=9)
    set_point(cloud, cfg, 0, 6, posC, 0.2, label=9)
    # camera looks away (yaw 180 deg about y) so that visibility / weight update leave the voxels alone
    q_back = np.array([0, 0, 1, 0], np.float32)
    m.update(depth, cloud, ORIGIN, q_back, stop_after="birth")
    o = m.dump_state()
    exp = _resample_expect(wA)
    assert exp is not None and exp.count(None) >= 1
    first_free = None
    for s, e in enumerate(exp, start=1):
        i = vA * S + s
        if e is None:
            if first_free is None:
                first_free = s
            else:
                assert o["status"][i] == INVALID and o["owner"][i] == 0xFFFF   # killed + erased from its owner set
        else:
            assert o["status"][i] == UPDATED and o["w"][i] == e
    i = vA * S + first_free                                                    # the retried birth took the first freed slot
    assert o["status"][i] == REGULAR_BORN and o["label"][i] == 9 and o["owner"][i] == 3 and o["ts"][i] == 2
    # B: everything UPDATED wiped, then the birth lands in slot 1
    assert o["status"][vB * S + 1] == REGULAR_BORN and np.all(o["status"][vB * S + 2:(vB + 1) * S] == INVALID)
    # C: untouched, birth lost
    assert np.all(o["status"][vC * S + 1:vC * S + 5] == UPDATED) and np.all(o["w"][vC * S + 1:vC * S + 5] == F(0.2))
    assert not np.any(o["label"][vC * S:(vC + 1) * S] == 9)


# ---------------------------------------------------------------------------------------------- (6) A10
def case_occupancy_codes(make):
    """determineIfVoxelOccupied / calculateWeightAndSemanticsInVoxel (operations.h:390-448, 623-639)."""
    cfg = K0
    S = 8
    st = empty_state(cfg)
    base = [(-3.1 + 0.5 * k, -3.1, -3.1) for k in range(8)]          # 8 voxels behind the camera
    v = [voxel_of(cfg, *p) for p in base]
    # v0: unknown (time particle stamp 0) although it holds a particle
    put(st, cfg, v[0], 1, base[0], 0.9)
    # v1: free: observed, weight below the threshold
    st["ts"][v[1] * S] = 1
    put(st, cfg, v[1], 1, base[1], 0.1, track=7, label=3)
    # v2: occupied; tie between track 9 (0.25+0.25) and track 4 (0.5): smaller id wins; label = last contributor's
    st["ts"][v[2] * S] = 1
    put(st, cfg, v[2], 1, base[2], 0.25, track=9, label=1)
    put(st, cfg, v[2], 2, base[2], 0.5, track=4, label=2)
    put(st, cfg, v[2], 3, base[2], 0.25, track=9, label=5)
    put(st, cfg, v[2], 5, base[2], 0.5, track=4, label=6)
    put(st, cfg, v[2], 6, base[2], 0.5, track=9, label=8)
    # v3: clamp: weight 3.0 counts fully in the sum, is written back as 1.0 and votes with 1.0
    st["ts"][v[3] * S] = 1
    put(st, cfg, v[3], 1, base[3], 3.0, track=11, label=4)
    put(st, cfg, v[3], 2, base[3], 0.7, track=12, label=5)
    put(st, cfg, v[3], 3, base[3], 0.6, track=12, label=5)
    # v4: low-weight UPDATED particle is culled (-> INVALID), REGULAR_BORN with the same weight is not
    st["ts"][v[4] * S] = 1
    put(st, cfg, v[4], 1, base[4], 0.04, UPDATED, track=20, label=1)
    put(st, cfg, v[4], 2, base[4], 0.04, REGULAR_BORN, track=21, label=2)
    # v5: guessed occupied: below threshold but guessed weight >= 0.05
    st["ts"][v[5] * S] = 1
    put(st, cfg, v[5], 1, base[5], 0.06, GUESSED_BORN, track=30, label=7)
    # v6: stale particle (stamp older than the slab stamp) is vacant; voxel itself stale -> unknown
    st["ts"][v[6] * S] = 1
    put(st, cfg, v[6], 1, base[6], 0.9, ts=1)
    # v7: only culled contributors -> sum above threshold is impossible here; check (0,0) labels on a free voxel
    st["ts"][v[7] * S] = 1
    m = fresh(make, st=st)
    sx, sy, sz = m.stamps()
    sx = sx.copy()
    sx[v[6] & 15] = 2                                               # x-slab of v6 recycled at frame 2 > its stamps
    st2 = m.dump_state()
    st2["ts"][v[6] * S] = 1
    m.set_stamps(sx, sy, sz)
    depth, cloud = blank_frame(cfg)
    m.update(depth, cloud, ORIGIN, IDENT_Q)
    r = m.voxels()
    o = m.dump_state()
    assert r["occ"][v[0]] == -1 and r["wsum"][v[0]] == F(-1) and r["track"][v[0]] == 0 and r["label"][v[0]] == 0
    assert r["occ"][v[1]] == 0 and r["wsum"][v[1]] == F(0.1) and r["track"][v[1]] == 7 and r["label"][v[1]] == 3
    assert r["occ"][v[2]] == 1 and r["wsum"][v[2]] == F(F(F(F(F(0.25) + F(0.5)) + F(0.25)) + F(0.5)) + F(0.5))
    assert r["track"][v[2]] == 4 and r["label"][v[2]] == 6          # 1.0 vs 1.0 tie -> track 4; its last label
    assert r["occ"][v[3]] == 1 and r["wsum"][v[3]] == F(F(F(3.0) + F(0.7)) + F(0.6))
    assert o["w"][v[3] * S + 1] == F(1.0)                           # clamp written back
    assert r["track"][v[3]] == 12 and r["label"][v[3]] == 5         # 1.3 beats the clamped 1.0
    assert r["occ"][v[4]] == 0 and o["status"][v[4] * S + 1] == INVALID and o["status"][v[4] * S + 2] == REGULAR_BORN
    assert r["wsum"][v[4]] == F(F(0.04) + F(0.04)) and r["track"][v[4]] == 21
    assert r["occ"][v[5]] == 2 and r["track"][v[5]] == 30 and r["label"][v[5]] == 7
    assert r["occ"][v[6]] == -1
    assert r["occ"][v[7]] == 0 and r["wsum"][v[7]] == 0 and r["track"][v[7]] == 0 and r["label"][v[7]] == 0


# ---------------------------------------------------------------------------------------------- (7) A6
def case_visibility(make):
    """getIdxOfVisibleParitlces (operations.h:1368-1431): binning, occlusion at 1.1*depth, far-depth reset,
    stale deletion, time-particle stamping of observed and of empty voxels."""
    cfg = K0
    S = 8
    W = cfg["width"]
    st = empty_state(cfg)
    pv, po, pf, ps = (0.1, 0.1, 2.1), (0.6, 0.1, 2.6), (-0.6, 0.1, 2.1), (0.1, 0.6, 2.1)
    vv, vo, vf, vs = (voxel_of(cfg, *p) for p in (pv, po, pf, ps))
    iv = put(st, cfg, vv, 1, pv, 0.4)            # visible: depth 2.0 at its pixel, 2.1 <= 2.2
    io = put(st, cfg, vo, 1, po, 0.4)            # occluded: 2.6 > 1.1 * 2.0
    i_f = put(st, cfg, vf, 1, pf, 0.4)           # pixel depth beyond depth_max -> weight reset to 0.05, not binned
    is_ = put(st, cfg, vs, 1, ps, 0.4, ts=1)     # stale: its y-slab was recycled at stamp 3
    m = fresh(make, st=st, gts=3)
    sx, sy, sz = m.stamps()
    sy = sy.copy()
    sy[(vs >> 4) & 15] = 3
    m.set_stamps(sx, sy, sz)
    depth, cloud = blank_frame(cfg, depth_value=2.0)

    def pixel(p):  # (fx*x + cx*z)/z truncated (operations.h:1279-1281)
        return int((F(20) * F(p[1]) + F(12) * F(p[2])) / F(p[2])), int((F(20) * F(p[0]) + F(16) * F(p[2])) / F(p[2]))

    rf, cf = pixel(pf)
    depth[rf * W + cf] = 50.0
    m.update(depth, cloud, ORIGIN, IDENT_Q, stop_after="visibility")
    o = m.dump_state()
    counts = m.bin_counts()
    rv, cv = pixel(pv)
    assert counts.sum() == 1 and counts[rv, cv] == 1 and list(m.bins()) == [iv]
    assert o["ts"][vv * S] == 4                                   # observed voxel stamped with the new frame (gts 3 -> 4)
    assert o["ts"][vo * S] == 0 and o["status"][io] == UPDATED    # occluded: nothing happens, voxel not observed
    assert o["w"][i_f] == F(0.05) and o["ts"][vf * S] == 4        # free-space reset, voxel observed
    assert o["status"][is_] == INVALID                            # stale particle deleted
    # an empty voxel in front of the surface is stamped when its min corner projects in front of the depth
    ve = voxel_of(cfg, 0.1, 0.1, 1.1)
    assert o["ts"][ve * S] == 4
    vb = voxel_of(cfg, 0.1, 0.1, 3.6)                             # behind the 2.0 m surface: stays unknown
    assert o["ts"][vb * S] == 0


# ---------------------------------------------------------------------------------------------- (9) A7
def case_weight_closed_form(make, pdf_table):
    """One particle, one valid pixel: w' = w * (P_d*g/(P_d*w*g + kappa) + 1 - P_d) with g = LUT product
    (semantic_dsp_map.h:1016-1035, 1085-1103), in float32 in the reference's operation order."""
    cfg = K0
    S = 8
    W = cfg["width"]
    st = empty_state(cfg)
    p = (0.1, 0.1, 2.1)
    v = voxel_of(cfg, *p)
    i = put(st, cfg, v, 1, p, 0.4, UPDATED, ts=1, track=65535, forget=2)
    m = fresh(make, st=st)
    depth, cloud = blank_frame(cfg, depth_value=2.0)
    row = int((F(20) * F(p[1]) + F(12) * F(p[2])) / F(p[2]))
    col = int((F(20) * F(p[0]) + F(16) * F(p[2])) / F(p[2]))
    obs = (0.15, 0.05, 2.0)
    sigma = F(0.2)
    set_point(cloud, cfg, row, col, obs, sigma)
    m.update(depth, cloud, ORIGIN, IDENT_Q, stop_after="weight")
    o = m.dump_state()

    def lut(x, mu):
        c = F(F(F(x) - F(mu)) / sigma)
        return pdf_table[int(F(F(c * F(1000)) + F(10000)))]

    g = F(F(lut(p[0], obs[0]) * lut(p[1], obs[1])) * lut(p[2], obs[2]))
    forget = F(2.5 ** (-2 / 1.0))                         # getForgettingFactor(2), basic_algorithms.h:32-48
    gk1 = F(g * forget)
    ck = F(F(0.4) * gk1)
    ck_kappa = F(F(ck * F(0.9)) + F(0.05))
    gk2 = F(g * forget)                                   # pass 2: same track -> no id transition, then forgetting
    acc = F(gk2 / ck_kappa)
    w_new = F(F(0.4) * F(F(F(acc * F(0.9)) + F(1.0)) - F(0.9)))
    assert o["w"][i] == w_new
    assert o["status"][i] == UPDATED and o["ts"][i] == 2
    assert o["forget"][i] == (0 if g > F(0.1) else 3)    # updated with the right id -> forget count reset
    assert m.ck_kappa()[row, col] == ck_kappa


# ---------------------------------------------------------------------------------------------- (2) A1
def case_ring_shift(make):
    """updateEgoCenterPos / updateRingbufferIndexParams (operations.h:68-96, 1111-1191): +k voxels on x stamps
    exactly k slabs, eq_steps = k mod N; particles in recycled slabs become vacant, others stay; a negative move
    stamps from the other end; a jump beyond a quarter of the axis is split but ends in the same state."""
    cfg = K0
    S = 8
    st = empty_state(cfg)
    keep = (3.1, 0.1, 0.1)       # map x index 14
    lose = (-3.9, 0.1, 0.1)      # map x index 0: first slab to be recycled when the ego moves +x
    vk, vl = voxel_of(cfg, *keep), voxel_of(cfg, *lose)
    put(st, cfg, vk, 1, keep, 0.9)
    put(st, cfg, vl, 1, lose, 0.9)
    st["ts"][vk * S] = 1
    st["ts"][vl * S] = 1
    m = fresh(make, st=st)
    depth, cloud = blank_frame(cfg)
    q_up = np.array([np.sqrt(0.5), np.sqrt(0.5), 0, 0], np.float32)   # look along -y/+y: nothing of interest in view
    m.update(depth, cloud, np.array([1.6, 0, 0], np.float32), q_up)   # +3 voxels (trunc(1.6/0.5) = 3)
    rs = m.ring_state()
    sx, sy, sz = m.stamps()
    assert rs["moved_steps"] == [3, 0, 0] and rs["eq_steps"] == [3, 0, 0]
    assert list(np.flatnonzero(sx)) == [0, 1, 2] and np.all(sx[:3] == 2) and not sy.any() and not sz.any()
    assert np.allclose(rs["map_center"], [1.5, 0, 0])
    r = m.voxels()
    assert r["occ"][vl] == -1                      # its slab was recycled: unknown again
    assert r["occ"][vk] == 1                       # retained: same storage slot, still occupied
    # negative move by 5 from there: stamps ring indices (N-1-i + eq) wrapped, i = 0..4
    m.update(depth, cloud, np.array([-1.2, 0, 0], np.float32), q_up)  # trunc(-1.2/0.5) = -2 -> new_moved = -5
    rs = m.ring_state()
    sx, _, _ = m.stamps()
    assert rs["moved_steps"] == [-2, 0, 0] and rs["eq_steps"] == [-2, 0, 0]
    want = sorted(((15 - i + 3) % 16) for i in range(5))
    assert sorted(np.flatnonzero(sx == 3)) == want
    # big jump on z: 7.3 m = 14 voxels > quarter axis (4 voxels = 2 m) -> split, final state as a direct computation
    m.update(depth, cloud, np.array([-1.2, 0, 7.3], np.float32), q_up)
    rs = m.ring_state()
    _, _, sz = m.stamps()
    assert rs["moved_steps"][2] == 14 and rs["eq_steps"][2] == 14
    assert np.count_nonzero(sz == 4) == 14


ALL_CASES = [case_first_vacant_slot_and_full_voxel, case_resample, case_occupancy_codes, case_visibility,
             case_ring_shift]
