O=gpurun_out/r04_race; mkdir -p $O; rm -f /tmp/g_ref.pt
one() { tag=$1; shift; echo "=== $tag [$@]"; env RACE_FRESH=0 RACE_REF=/tmp/g_ref.pt "$@" timeout 600 python tools/race_probe.py $tag 30 2>&1 | grep "differs\|done\|Error\|first gradient" | cut -c1-300 | head -5; }
one default X=1
one gnbwd0 FAVAE_GNBWD_FUSE=0
one dycs0 FAVAE_DYCS_FUSE=0
one defer0 FAVAE_DEFER_REDUCE=0
one side0 FAVAE_WGRAD_STREAM=0
one all0 FAVAE_GNBWD_FUSE=0 FAVAE_DYCS_FUSE=0 FAVAE_DEFER_REDUCE=0 FAVAE_WGRAD_STREAM=0
