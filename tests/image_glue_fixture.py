"""Loader of tests/golden/image_glue_vectors.npz (shared by the CPU and GPU tests):
rebuilds the img_metas dicts the reference functions were run on."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                    "image_glue_vectors.npz")


class Fixture:
    def __init__(self):
        g = self.g = np.load(GOLD)
        d = [int(v) for v in g["meta_dims"]]
        self.B, self.cams, self.c_img, self.c_out, self.H, self.W = d[:6]
        self.scales = d[6:]
        self.metas = []
        for b in range(self.B):
            info = dict(fg_pixels=[g[f"pix_{b}_{j}"] for j in range(self.cams)],
                        fg_points=[g[f"pts_{b}_{j}"] for j in range(self.cams)],
                        fg_real_pixels=[g[f"real_{b}_{j}"] for j in range(self.cams)])
            self.metas.append(dict(foreground2D_info=info,
                                   lidar2img=[g[f"l2i_{b}_{j}"] for j in range(self.cams)],
                                   input_shape=(self.H, self.W), pad_shape=(self.H, self.W, 3)))
        self.feats = [g[f"feat_{i}"] for i in range(len(self.scales))]
        self.comp = [g[f"comp_{i}"] for i in range(len(self.scales))]
        # get_foreground2D ran on [comp0, comp0, comp1, comp2]
        self.fg_inputs = [self.comp[0]] + self.comp
        self.fg = [[g[f"fg_{i}_{b}"] for b in range(self.B)] for i in range(len(self.fg_inputs))]

    def state_dict(self, prefix):
        import torch
        n = len("w_" + prefix)
        return {k[n:]: torch.from_numpy(self.g[k]) for k in self.g.files
                if k.startswith("w_" + prefix)}
