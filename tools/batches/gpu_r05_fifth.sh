#!/bin/bash
# GPU box (round 5): what the dolly fill's loop does with and without the cooperative creeping rays; the density sweep; GPU suite
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_fifth
mkdir -p $O
cd $R
echo "== fill stats, coop 0"; EXTRA="-DKBE_FILL_COOP_LANES=0" timeout 900 python tools/fill_stats.py 2>&1 | tee $O/fill_stats_coop0.txt | tail -6
echo "== fill stats, coop 4"; timeout 900 python tools/fill_stats.py 2>&1 | tee $O/fill_stats_coop4.txt | tail -6
echo "== density sweep"; timeout 1200 python tools/density_sweep.py 2>&1 | tee $O/density_sweep.txt | grep "points per pixel"
echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
