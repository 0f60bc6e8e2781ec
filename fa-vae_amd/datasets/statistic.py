"""Normalisation constants the reference scripts import by name (reference datasets/statistic.py:2-7)."""
_HALF = 0.5

# images are mapped to [-1, 1] before the encoder / LPIPS: (x - 0.5) / 0.5 per channel
mean = [_HALF] * 3
std = [_HALF] * 3

# OpenAI CLIP image preprocessing (stage-2 scripts only)
clip_mean = [0.48145466, 0.4578275, 0.40821073]
clip_std = [0.26862954, 0.26130258, 0.27577711]
