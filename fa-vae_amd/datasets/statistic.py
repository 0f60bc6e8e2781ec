# normalisation constants the reference training script imports (reference datasets/statistic.py:2-7)
mean = [0.5, 0.5, 0.5]
std = [0.5, 0.5, 0.5]

# CLIP preprocessing constants (reference datasets/statistic.py:6-7; used by the stage-2 scripts only)
clip_mean = [0.48145466, 0.4578275, 0.40821073]
clip_std = [0.26862954, 0.26130258, 0.27577711]
