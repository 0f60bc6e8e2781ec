# normalisation constants the reference training script imports (reference datasets/statistic.py:2-7)
mean = [0.5, 0.5, 0.5]
std = [0.5, 0.5, 0.5]
