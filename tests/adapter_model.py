"""TEST INFRASTRUCTURE: a Python restatement of what the reference's SemanticDSPMap::update does around the hot path
(/root/reference/include/semantic_dsp_map.h:170-251, 588-736, 1239-1376), put together from the oracle's pieces:

  track-id reallocation (:179-186)  ->  object layer (oracle/object_layer.py; :189-191, 588-736, floating objects from the
  owner sets' keys :712-736)  ->  generateLabeledPointCloud (oracle.generate_cloud_ex; utils/pointcloud_tools.h:88-310,
  the ZED2 boxes of :174-196 computed here)  ->  subObjectLevelUpdate (oracle.update)  ->  getOccupancyResult's clouds in
  storage order (:1244-1376) coloured by oracle/colour.py and packed as pcl::PointXYZRGB.

tests/test_adapter_parity.py holds the product's C++ class (include/semantic_dsp_map.h on libsdm_hip) to this model byte
for byte; nothing of the product imports it."""
import numpy as np

from oracle import colour as col
from oracle import object_layer as ol
from oracle import oracle as orc
from semantic_dsp_map_amd import binding, synth

# utils/data_base.h:108-232 (what SemanticDSPMap::defaultLabelTables holds when the node sets no tables)
LABEL_IDS = {"Background": 0, "Terrain": 2, "Sky": 3, "Tree": 4, "Vegetation": 5, "Building": 6, "Road": 7, "GuardRail": 8,
             "TrafficSign": 9, "TrafficLight": 10, "Pole": 11, "Misc": 12, "Truck": 13, "Car": 14, "Person": 15}
STATIC_NAMES = ["Background", "Terrain", "Sky", "Tree", "Vegetation", "Building", "Road", "GuardRail", "TrafficSign",
                "TrafficLight", "Pole", "Misc"]
LABEL_BGR = np.zeros((256, 3), np.uint8)
for _l, _c in {0: (0, 0, 0), 2: (200, 0, 210), 3: (255, 200, 90), 4: (0, 199, 0), 5: (0, 240, 90), 6: (140, 140, 140),
               7: (100, 60, 100), 8: (255, 100, 250), 9: (0, 255, 255), 10: (0, 200, 200), 11: (0, 130, 255), 12: (80, 80, 80),
               13: (60, 60, 160), 14: (80, 127, 255), 15: (139, 139, 0)}.items():
    LABEL_BGR[_l] = _c
MAX_MOVABLE = 65523


def default_perm():
    """the adapter's fixed permutation (the reference shuffles with an unseeded generator, semantic_dsp_map.h:45-48)"""
    perm = list(range(256))
    s = 12345
    for i in range(255, 0, -1):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        j = (s >> 8) % (i + 1)
        perm[i], perm[j] = perm[j], perm[i]
    return np.array(perm, np.uint8)


def label_to_instance():
    tab = np.full(256, 65535, np.uint16)
    for i, name in enumerate(STATIC_NAMES):  # static instance 65535 - i <-> label of the i-th static class
        tab[LABEL_IDS[name]] = 65535 - i
    return tab


class AdapterModel:
    def __init__(self, preset, params, noise, bayes=(0.1, 0.69, 0.1, 0.15), object_seed=20250217, evaluation_format=False):
        self.p = dict(preset)
        self.params = dict(params)
        self.cfg = dict(x_n=preset["x_n"], y_n=preset["y_n"], z_n=preset["z_n"], p_n=preset["p_n"], voxel_size=preset["voxel_size"],
                        fx=preset["fx"], fy=preset["fy"], cx=preset["cx"], cy=preset["cy"], width=preset["width"],
                        height=preset["height"], depth_min=preset["depth_min"], depth_max=preset["depth_max"],
                        window_half=preset["window_half"], max_movable_track=MAX_MOVABLE, bin_order=1)
        self.o = orc.OracleMap(self.cfg, params, noise)
        self.gts = 0
        self.evaluation_format = evaluation_format
        self.perm = default_perm()
        self.l2i = label_to_instance()
        self.layer = None
        if preset["consider_instance"]:
            n_big = max(preset["x_n"], preset["y_n"], preset["z_n"])
            self.layer = ol.ObjectLayer(dict(
                mode=preset["object_mode"], max_movable_instance_id=MAX_MOVABLE, movement_distance_threshold=bayes[0],
                movement_probability_threshold=bayes[1], movement_increment=bayes[2], movement_decrement=bayes[3],
                map_half_size_scaled=float(np.float32(preset["voxel_size"])) * float(1 << (n_big - 1)) * 1.2,
                fx=float(np.float32(preset["fx"])), fy=float(np.float32(preset["fy"])), cx=float(np.float32(preset["cx"])),
                cy=float(np.float32(preset["cy"])), image_width=preset["width"], image_height=preset["height"], seed=object_seed))

    def tracks_with_particles(self):
        own = self.o.dump_state()["owner"]
        return sorted(int(x) for x in np.unique(own[own != 0xFFFF]))

    def update(self, depth, seg, cam_pos, cam_q, get_freespace=False, time_stamp=0.0):
        """seg: list of dicts track_id, label (str), kpts_current (n x 3), kpts_previous (n x 3 or None), mask (H x W uint8).
        Returns (occupied, free) as POINT_XYZRGB arrays; free is None unless asked for."""
        p = self.p
        self.gts += 1
        seg = [dict(s) for s in seg]
        for s in seg:  # :179-186
            if s["label"] != "static" and s["track_id"] > MAX_MOVABLE:
                s["track_id"] = s["track_id"] % MAX_MOVABLE
        moves, removals = [], []
        if self.layer is not None:
            obs = []
            for s in seg:
                cur = np.asarray(s["kpts_current"], np.float64).reshape(-1, 3)
                prev = s.get("kpts_previous")
                prev = None if prev is None else np.asarray(prev, np.float64).reshape(-1, 3)
                if prev is not None and (len(prev) != len(cur) or len(cur) == 0):
                    prev = None
                obs.append(dict(track_id=s["track_id"], label_id=LABEL_IDS.get(s["label"], -1), is_static=s["label"] == "static",
                                kpts_current=cur, kpts_previous=prev))
            present = self.tracks_with_particles()
            self.layer.update(obs, np.asarray(cam_pos, np.float64), np.asarray(cam_q, np.float64), time_stamp, self.gts)
            moves, removals = self.layer.collect(self.gts, self.params["max_obersevation_lost_time"], present)
        # packRawInputs: the first "static" entry, then one mask per other entry in order
        static_mask = None
        for s in seg:
            if s["label"] == "static":
                static_mask = np.asarray(s["mask"], np.uint8)
                break
        objects, boxes = [], []
        if p["consider_instance"]:
            for s in seg:
                if s["label"] == "static":
                    continue
                objects.append((s["track_id"], LABEL_IDS.get(s["label"], 0), np.asarray(s["mask"], np.uint8)))
                if p["zed2_filters"]:  # pointcloud_tools.h:174-196 (the maximum starts at the smallest positive double there)
                    lo = np.full(3, np.finfo(np.float64).max)
                    hi = np.full(3, np.finfo(np.float64).tiny)
                    for k in np.asarray(s["kpts_current"], np.float64).reshape(-1, 3):
                        lo = np.minimum(lo, k)
                        hi = np.maximum(hi, k)
                    boxes.append([lo[0] - 1.0, hi[0] + 1.0, lo[1] - 1.0, hi[1] + 1.0, lo[2] - 1.0, hi[2] + 1.0])
        sky = -1
        if p["zed2_filters"] and "Sky" in LABEL_IDS:
            sky = int(self.l2i[LABEL_IDS["Sky"]])
        src = (p["src_width"], p["src_height"]) if p["src_width"] > 0 else None
        cloud, depth_small = self.o.generate_cloud_ex(depth, static_mask, self.l2i, objects, cam_pos, cam_q, p["consider_instance"],
                                                      src_size=src, rescale=p["rescale"], sky_instance=sky,
                                                      object_bbox=np.array(boxes, np.float64) if boxes else None)
        mv = np.zeros(len(moves), synth.OBJECT_MOVE)
        for i, (trk, T) in enumerate(moves):
            mv[i]["track_id"], mv[i]["T"] = trk, np.asarray(T, np.float32).reshape(16)
        self.o.update(depth_small.reshape(p["height"], p["width"]), cloud, np.asarray(cam_pos, np.float32), np.asarray(cam_q, np.float32),
                      mv, list(removals) if removals else None)
        occ = self.emit(False)
        return occ, (self.emit(True) if get_freespace else None)

    def emit(self, free):
        vox = self.o.voxels()
        sel = np.flatnonzero(vox["occ"] == 0) if free else np.flatnonzero(vox["occ"] > 0)  # storage order
        out = np.zeros(sel.size, binding.POINT_XYZRGB)
        pos = np.array([self.o.voxel_to_pos(int(v)) for v in sel], np.float32).reshape(-1, 3)
        out["x"], out["y"], out["z"] = pos[:, 0], pos[:, 1], pos[:, 2]
        out["one"] = 1.0
        out["a"] = 255
        if free:  # :1371-1373
            out["g"] = 255
            return out
        oof = np.array([not self.o.point_in_frustum(*pos[k]) for k in range(sel.size)], bool)
        rgb = col.colour_points(pos[:, 2], pos[:, 1], vox["track"][sel], vox["label"][sel], vox["occ"][sel], oof, LABEL_BGR, self.perm,
                                LABEL_IDS["Background"], MAX_MOVABLE, colour_by_label=not self.p["consider_instance"],
                                jet_axis=1 if self.p["zed2_filters"] else 0, evaluation_format=self.evaluation_format)
        out["r"], out["g"], out["b"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
        return out
