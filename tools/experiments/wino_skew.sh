cd $GRAFT_REPO_ROOT
for k in 0 1 3 6; do echo "skew $k"; FAVAE_WINO_SKEW=$k timeout 300 python tools/conv_bench.py 2>&1 | grep "k3" | cut -c1-120; done
