for a in wgrad conv; do python tools/experiments/apply_race.py 200 $a 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "--- wgrad = three-tap kernel (FAVAE_WGRAD_NINE=0)"; FAVAE_WGRAD_NINE=0 python tools/experiments/apply_race.py 200 wgrad 2>&1 | grep -v amdgpu.ids | tail -3
echo "--- wgrad = per-tap kernel (FAVAE_WGRAD_NINE=0 FAVAE_WGRAD_ROW3=0)"; FAVAE_WGRAD_NINE=0 FAVAE_WGRAD_ROW3=0 python tools/experiments/apply_race.py 200 wgrad 2>&1 | grep -v amdgpu.ids | tail -3
