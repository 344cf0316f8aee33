"""Module-level constants mirroring the reference's ``config.py:7-95`` (the whole flag system there).

Dataset paths are not reproduced (the reference ships ``***`` placeholders); everything on the hot path is.
"""
from enum import Enum


class Modality(Enum):          # config.py:7-11
    ct = 0
    t1in = 1
    t1out = 2
    t2 = 3


seed = 2020                    # config.py:23
n_modal = len(Modality.__members__)
n_label = 4                    # config.py:26 (CHAOS: 4 organs + background)

num_iter_per_epoch = 150       # config.py:29
max_epoch = 200
exp_alpha = 1.0
weight_dc = 0.5                # config.py:32-33
weight_ce = 0.5

img_channels = 1
base_width = 16
input_size = 256               # config.py:50
batch_size = 8                 # config.py:56
num_workers = 6

lr = 1e-2                      # config.py:73-74
weight_decay = 1e-3
nce_layers = [5]               # config.py:77

expr_root = "smsut_out"        # the reference's placeholder is '***/bimod-out' (config.py:46)
base_root = None               # processed PNG dataset root ('***/bimod' upstream, config.py:44); None -> synthetic slices
split_yaml = "semi-1910.yaml"  # config.py:54
data_aug = dict(               # config.py:60-71
    rotate=True, rotate_degrees=15,
    resizeCrop=True, resizeCrop_size=input_size,
    elasticDeform=True, elasticDeform_sigmas=(9., 13.), elasticDeform_points=3,
    colorJitter=False, gammaCorrect=False, gammaCorrect_gammas=(0.7, 1.5),
)
