"""Virtual-point on-disk format and its loaders: the data format on
the input side of the hot path.

Reference: mmdet3d/datasets/pipelines/my_loading_multi_proj.py -- `LoadForeground2D`
(:14-161) and `LoadForeground2DFromMultiSweeps` (:163-338).  One file per LiDAR sweep,
`<root>/FOREGROUND_MIXED_6NN_WITH_DEPTH/<lidar file name>.pkl.npy`, a pickled dict
(np.save of a dict, read back with np.load(allow_pickle=True).item()):

    virtual_pixel_indices / real_pixel_indices : per camera [n, 3 + 11]
        (x, y, depth, 10 one-hot class scores + 1 score) in the ORIGINAL image scale
    virtual_points / real_points               : per camera [n, 3] (LiDAR frame)

`__call__(results)` has the reference's pipeline semantics (same keys in, same keys
out): results['foreground2D_info'] = dict(fg_pixels, fg_points, fg_real_pixels,
fg_real_points), one array per camera, fg_points = [xyz | 11 labels | dt] (15 columns:
what get_foreground2D concatenates image features to).  Pure numpy -- this is host-side
data-loader work; `msmdfusion_amd.image_glue.pack_foreground` is where it meets the
device.
"""
import os

import numpy as np

FOREGROUND_DIR = "FOREGROUND_MIXED_6NN_WITH_DEPTH"


class LidarPointsView:
    """The slice of mmdet3d's LiDARPoints the loaders and the hot path use: `.tensor`
    (float32 [n, dim]; MSMDFusion.py:221 reads it), `points_dim`, `new_point`, `cat`,
    row / column indexing (core/points/base_points.py)."""

    def __init__(self, array, points_dim=None):
        import torch
        self.tensor = torch.as_tensor(np.asarray(array), dtype=torch.float32)
        self.points_dim = self.tensor.shape[-1] if points_dim is None else points_dim

    def __len__(self):
        return self.tensor.shape[0]

    def new_point(self, data):
        return LidarPointsView(data)

    @staticmethod
    def cat(points_list):
        import torch
        return LidarPointsView(torch.cat([p.tensor for p in points_list], 0).numpy())

    def __getitem__(self, item):
        import torch
        if isinstance(item, np.ndarray) and item.dtype == np.bool_:
            item = torch.from_numpy(item)
        return LidarPointsView(self.tensor[item].numpy())


def foreground_path(lidar_path, directory=FOREGROUND_DIR, suffix=".pkl.npy"):
    """:127-128: the sweep's directory is swapped for the foreground directory.
    (The reference joins the path tokens with os.path.join(*tokens), which drops a
    leading '/': it notes the bug itself; absolute paths are kept absolute here.)"""
    head, name = os.path.split(lidar_path)
    root = os.path.dirname(head)
    return os.path.join(root, directory, name + suffix)


def save_foreground(path, virtual_pixel_indices, real_pixel_indices, virtual_points,
                    real_points):
    """Writer of the format (the reference only reads it; its files come from the
    authors' offline virtual-point generator)."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    payload = dict(virtual_pixel_indices=list(virtual_pixel_indices),
                   real_pixel_indices=list(real_pixel_indices),
                   virtual_points=list(virtual_points), real_points=list(real_points))
    with open(path, "wb") as f:          # (np.save would append '.npy' to the name)
        np.save(f, np.array(payload, dtype=object), allow_pickle=True)


def load_foreground(path):
    return np.load(path, allow_pickle=True).item()


def _with_labels(points, pixel_indices):
    # :75-77 "append label after xyz": the last 11 columns of the pixel record
    return np.concatenate((points, pixel_indices[:, -11:]), axis=1) \
        if points.shape[1] == 3 else points


def organize(fg_info, dt=0.0, dt_real=0.0):
    """`_organize` of both loaders (:49-97, :171-224): per camera, virtual then real
    rows; pixels keep (x, y, depth); points get their 11 label columns and a time
    column.  Like the reference, the label columns are written back into fg_info."""
    cams = len(fg_info["virtual_pixel_indices"])
    out = dict(fg_pixels=[], fg_points=[], fg_real_pixels=[], fg_real_points=[])
    for i in range(cams):
        vpix, rpix = fg_info["virtual_pixel_indices"][i], fg_info["real_pixel_indices"][i]
        fg_info["virtual_points"][i] = _with_labels(fg_info["virtual_points"][i], vpix)
        fg_info["real_points"][i] = _with_labels(fg_info["real_points"][i], rpix)
        vpts, rpts = fg_info["virtual_points"][i], fg_info["real_points"][i]
        pts = np.concatenate((vpts, rpts), axis=0)
        out["fg_pixels"].append(np.concatenate((vpix[:, :3], rpix[:, :3]), axis=0))
        out["fg_points"].append(np.concatenate((pts, np.full((pts.shape[0], 1), dt)), axis=1))
        out["fg_real_pixels"].append(rpix[:, :3])
        out["fg_real_points"].append(
            np.concatenate((rpts, np.full((rpts.shape[0], 1), dt_real)), axis=1))
    return out


class LoadForeground2D:
    """:14-161.  nuScenes: arrays per camera (the multi-sweep loader wraps them
    later); KITTI: one pseudo-camera, `virtual_1NN/<frame>.npy`, wrapped at once."""

    def __init__(self, dataset="NuScenesDataset", **kwargs):
        self.dataset = dataset

    def __call__(self, results):
        if self.dataset == "NuScenesDataset":
            fg_info = load_foreground(foreground_path(results["pts_filename"]))
            results["foreground2D_info"] = organize(fg_info)
            return results
        if self.dataset == "KittiDataset":
            head, name = os.path.split(results["pts_filename"])
            path = os.path.join(os.path.dirname(head), "virtual_1NN",
                                name.split(".")[0] + ".npy")
            fg_info = load_foreground(path)
            if len(fg_info.keys()) == 4:                                   # :99-103
                pix = np.concatenate((fg_info["virtual_pixel_indices"],
                                      fg_info["real_pixel_indices"]), axis=0)
                pts = np.concatenate((fg_info["virtual_points"], fg_info["real_points"]), axis=0)
            else:
                pix, pts = np.zeros((0, 2)), np.zeros((0, 6))
            results["foreground2D_info"] = dict(fg_pixels=[pix],
                                                fg_points=[LidarPointsView(pts)])
            return results
        raise NotImplementedError(
            "foreground2D info of {} dataset is unavailable!".format(self.dataset))


class LoadForeground2DFromMultiSweeps:
    """:163-338.  Adds the foreground points of up to `sweeps_num` earlier sweeps to the
    key frame's (results['foreground2D_info'] from LoadForeground2D): sweep points are
    moved into the key frame (p @ R^T + t with the sweep's sensor2lidar_*), their time
    column is ts - sweep_ts (seconds).  Kept as the reference has them: the REAL points'
    time column is ts - sweep_ts / 1e-6 (:216; not used downstream), sweeps without a
    foreground file are skipped, pixels of sweeps are appended unaligned (:251).
    `test_mode` is read by the reference but never set in its __init__ (:301);
    here it is a constructor argument."""

    def __init__(self, dataset="NuScenesDataset", sweeps_num=10, test_mode=False):
        self.dataset = dataset
        self.sweeps_num = sweeps_num
        self.test_mode = test_mode

    @staticmethod
    def merge_sweep(fg_info, sweep_info, sweep):
        """`_merge_sweeps` (:226-283), camera by camera."""
        if len(sweep_info["fg_points"]) != len(fg_info["fg_points"]):
            return fg_info                       # (the reference prints a banner and moves on)
        rot_t = np.asarray(sweep["sensor2lidar_rotation"]).T
        trans = np.asarray(sweep["sensor2lidar_translation"])
        for cam in range(len(fg_info["fg_pixels"])):
            for pix_key, pts_key in (("fg_pixels", "fg_points"),
                                     ("fg_real_pixels", "fg_real_points")):
                pts = sweep_info[pts_key][cam]
                pts[:, :3] = pts[:, :3] @ rot_t
                pts[:, :3] = pts[:, :3] + trans
                fg_info[pix_key][cam] = np.concatenate(
                    (fg_info[pix_key][cam], sweep_info[pix_key][cam]), axis=0)
                fg_info[pts_key][cam] = np.concatenate((fg_info[pts_key][cam], pts), axis=0)
        return fg_info

    def __call__(self, results):
        if self.dataset != "NuScenesDataset":
            return None                          # (the reference falls off the end: None)
        fg_info = results["foreground2D_info"]
        n = len(results["sweeps"])
        if n <= self.sweeps_num:
            choices = np.arange(n)
        elif self.test_mode:
            choices = np.arange(self.sweeps_num)
        else:
            choices = np.random.choice(n, self.sweeps_num, replace=False)
        ts = results["timestamp"]
        for idx in choices:
            sweep = results["sweeps"][idx]
            path = foreground_path(sweep["data_path"])
            if not os.path.exists(path):
                continue
            sweep_ts = sweep["timestamp"] / 1e6
            sweep_info = organize(load_foreground(path), dt=ts - sweep_ts,
                                  dt_real=ts - sweep_ts / 1e-6)
            fg_info = self.merge_sweep(fg_info, sweep_info, sweep)
        fg_info["fg_points"] = [LidarPointsView(p, p.shape[-1]) for p in fg_info["fg_points"]]
        results["foreground2D_info"] = fg_info
        return results


def read_points(path, load_dim):
    """LoadPointsFromFile._load_points / LoadPointsFromMultiSweeps._load_points
    (loading.py:541-561): raw float32 records (.bin) or an .npy array."""
    pts = np.load(path) if path.endswith(".npy") else np.fromfile(path, dtype=np.float32)
    return np.copy(pts).reshape(-1, load_dim)


class LoadPointsFromFile:
    """loading.py `LoadPointsFromFile` for LiDAR coordinates: results['points'] =
    the file's first `use_dim` columns (an int n means range(n))."""

    def __init__(self, coord_type="LIDAR", load_dim=6, use_dim=(0, 1, 2), shift_height=False,
                 use_color=False, **kwargs):
        if coord_type != "LIDAR" or shift_height or use_color:
            raise NotImplementedError("only plain LIDAR point files are built")
        self.load_dim = load_dim
        self.use_dim = list(range(use_dim)) if isinstance(use_dim, int) else list(use_dim)

    def __call__(self, results):
        pts = read_points(results["pts_filename"], self.load_dim)[:, self.use_dim]
        results["points"] = LidarPointsView(pts)
        return results


class LoadPointsFromMultiSweeps:
    """loading.py:503-636: the key frame's points (time column zeroed) followed by up to
    `sweeps_num` earlier sweeps, each moved into the key frame's coordinates
    (p @ R^T + t), optionally without the points within `radius` of the sensor in x AND y,
    time column = ts - sweep_ts; finally the `use_dim` columns."""

    def __init__(self, sweeps_num=10, load_dim=5, use_dim=(0, 1, 2, 4), pad_empty_sweeps=False,
                 remove_close=False, test_mode=False, **kwargs):
        self.sweeps_num, self.load_dim, self.use_dim = sweeps_num, load_dim, list(use_dim)
        self.pad_empty_sweeps, self.remove_close, self.test_mode = \
            pad_empty_sweeps, remove_close, test_mode

    @staticmethod
    def _remove_close(points, radius=1.0):
        arr = points if isinstance(points, np.ndarray) else points.tensor.numpy()
        close = (np.abs(arr[:, 0]) < radius) & (np.abs(arr[:, 1]) < radius)
        return points[~close]

    def __call__(self, results):
        points = results["points"]
        points.tensor[:, 4] = 0
        sweeps = [points]
        ts = results["timestamp"]
        if self.pad_empty_sweeps and len(results["sweeps"]) == 0:
            for _ in range(self.sweeps_num):
                sweeps.append(self._remove_close(points) if self.remove_close else points)
        else:
            n = len(results["sweeps"])
            if n <= self.sweeps_num:
                choices = np.arange(n)
            elif self.test_mode:
                choices = np.arange(self.sweeps_num)
            else:
                choices = np.random.choice(n, self.sweeps_num, replace=False)
            for idx in choices:
                sweep = results["sweeps"][idx]
                pts = read_points(sweep["data_path"], self.load_dim)
                if self.remove_close:
                    pts = self._remove_close(pts)
                pts[:, :3] = pts[:, :3] @ np.asarray(sweep["sensor2lidar_rotation"]).T
                pts[:, :3] += np.asarray(sweep["sensor2lidar_translation"])
                pts[:, 4] = ts - sweep["timestamp"] / 1e6
                sweeps.append(points.new_point(pts))
        points = points.cat(sweeps)
        results["points"] = points[:, self.use_dim]
        return results
