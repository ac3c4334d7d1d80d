"""Golden fixture of the object layer: tests/golden/objects_clips.npz, generated from the numpy
restatement oracle/object_layer.py.

Run from the repository root:  python tests/golden/make_golden_objects.py
The reference ships no vectors for this code and its RANSAC is unseeded, so this pins the restatement (and the sampler
stream) against regressions and gives the product expected outputs that do not need the restatement at run time.  The
fixture holds the INPUTS (keypoints, poses, time stamps, as flat arrays) next to the expected outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import object_layer as ref  # noqa: E402

N_FRAMES, MAX_LOST = 24, 4


def config(mode):
    return dict(mode=mode, max_movable_instance_id=65000, movement_distance_threshold=0.1, movement_probability_threshold=0.69,
                movement_increment=0.2, movement_decrement=0.1, map_half_size_scaled=0.2 * 128 * 1.2,
                fx=725.0087, fy=725.0087, cx=620.5, cy=187.0, image_width=1242, image_height=375, seed=20250217)


def make_inputs(mode):
    """One seeded clip per mode: five objects (movers, a parked one, one with an unknown label, one far away), keypoint
    dropouts, garbage matches, a gap in the time stamps.  Returns flat arrays."""
    rng = np.random.default_rng(4242 + mode)
    n_k = 4 if mode in (0, 3) else 9
    base = {tid: rng.normal(size=(n_k, 3)) * np.array([1.0, 0.5, 2.0]) + np.array([tid * 2.5 - 8, 0.5, 9.0 + tid]) for tid in range(1, 6)}
    base[5] = base[5] + np.array([0, 0, 70.0])
    vel = {1: np.array([0.5, 0, 0.1]), 2: np.zeros(3), 3: np.array([-0.3, 0, 0.4]), 4: np.array([0.2, 0, -0.2]), 5: np.array([0.1, 0, 0])}
    rows, kp_cur, kp_prev, poses = [], [], [], []
    ts = 0.0
    for t in range(N_FRAMES):
        ts += 0.1 if t != 15 else 2.5
        poses.append([0.05 * t, 0, 0.1 * t, np.cos(0.01 * t), 0, np.sin(0.01 * t), 0, ts])
        for tid in base:
            prev = base[tid].copy()
            base[tid] = prev + vel[tid]
            if (tid == 3 and 8 <= t < 14) or rng.random() < 0.1:
                continue
            n = n_k if rng.random() > 0.15 else 2
            cur = base[tid][:n] + rng.normal(size=(n, 3)) * 0.01
            if rng.random() < 0.1:
                cur = cur + rng.normal(size=cur.shape) * 2
            rows.append([t, tid, -1 if tid == 4 and t < 6 else 15, 0, n, len(kp_cur)])
            kp_cur += cur.tolist()
            kp_prev += prev[:n].tolist()
        rows.append([t, 65535, 2, 1, 0, len(kp_cur)])
    return (np.array(rows, np.int64), np.array(kp_cur, np.float64).reshape(-1, 3), np.array(kp_prev, np.float64).reshape(-1, 3),
            np.array(poses, np.float64))


def frames_of(rows, kp_cur, kp_prev, poses):
    frames = []
    for t in range(len(poses)):
        obs = []
        for r in rows[rows[:, 0] == t]:
            _, tid, label, static, n, off = (int(x) for x in r)
            obs.append(dict(track_id=tid, label_id=label, is_static=bool(static), kpts_current=kp_cur[off:off + n],
                            kpts_previous=kp_prev[off:off + n] if n else None))
        frames.append((obs, poses[t, 0:3], poses[t, 3:7], float(poses[t, 7])))
    return frames


def present_at(t):
    return (1, 4, 5, 77) if t % 6 == 2 else ()


def run(layer, frames):
    """-> per frame move ids, move matrices (float32), removals, flattened with frame indices."""
    mv_rows, mv_T, rm_rows = [], [], []
    for t, (obs, pos, q, ts) in enumerate(frames):
        layer.update(obs, pos, q, ts, t + 1)
        moves, removals = layer.collect(t + 1, MAX_LOST, present_at(t))
        for tid, T in moves:
            mv_rows.append([t, tid])
            mv_T.append(np.asarray(T, np.float32).reshape(16))
        rm_rows += [[t, r] for r in removals]
    return (np.array(mv_rows, np.int64).reshape(-1, 2), np.array(mv_T, np.float32).reshape(-1, 16),
            np.array(rm_rows, np.int64).reshape(-1, 2))


def build():
    data = {}
    for mode in range(4):
        rows, kc, kp, poses = make_inputs(mode)
        mv, T, rm = run(ref.ObjectLayer(config(mode)), frames_of(rows, kc, kp, poses))
        for k, v in (("rows", rows), ("kp_cur", kc), ("kp_prev", kp), ("poses", poses), ("moves", mv), ("T", T), ("removals", rm)):
            data["m%d_%s" % (mode, k)] = v
    return data


if __name__ == "__main__":
    d = build()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "objects_clips.npz"), **d)
    for mode in range(4):
        print("mode", mode, len(d["m%d_moves" % mode]), "moves,", len(d["m%d_removals" % mode]), "removals")
