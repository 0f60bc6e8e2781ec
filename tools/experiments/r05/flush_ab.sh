# how many weight-gradient layers share one grouped slab reduction (FAVAE_FLUSH_EVERY, default 12): fewer = the slabs are reduced while they may still
# sit in the 256 MB Infinity Cache, more launches
AB_STEPS=8 AB_TOP=0 bash tools/ab_multi.sh r05_flushab "FAVAE_FLUSH_EVERY=12" "FAVAE_FLUSH_EVERY=4" "FAVAE_FLUSH_EVERY=2" "FAVAE_FLUSH_EVERY=1" "FAVAE_FLUSH_EVERY=12" "FAVAE_FLUSH_EVERY=4" "FAVAE_FLUSH_EVERY=2" "FAVAE_FLUSH_EVERY=1"
