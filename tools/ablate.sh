#!/bin/bash
for a in 0 1 2 3; do echo "ABLATE=$a"; FAVAE_B6_ABLATE=$a python tools/conv_bench.py 32 2>/dev/null | head -2; done
