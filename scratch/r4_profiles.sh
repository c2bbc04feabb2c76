#!/bin/bash
# Round 4's committed evidence in one GPU call: C3 bench + rocprofv3 stats + PMC passes, per-kernel clock counters, the C2
# launch chain, C4 counters.  usage (gpurun, repo root): bash scratch/r4_profiles.sh
bash scratch/profile_round.sh r04 2>&1 | tail -3
bash scratch/pmc_clock.sh r04clock > gpurun_out/r04clock.txt 2>&1; rm -rf gpurun_out/r04clock/*/
bash scratch/c2_round.sh r04c2 2>&1 | tail -2
python3 scratch/c2_summary.py gpurun_out/r04c2 > gpurun_out/r04c2/summary.txt 2>&1
bash scratch/pmc_cfg.sh r04c4 c4 > gpurun_out/r04c4.txt 2>&1; rm -rf gpurun_out/r04c4/*/
# keep what summarize_profile.py needs, drop the bulky raw traces
find gpurun_out/r04 gpurun_out/r04c2 -name "*_agent_info.csv" -delete; find gpurun_out/r04c2 -name "c2_kernel_trace.csv" -size +20M -delete
du -sh gpurun_out/r04 gpurun_out/r04c2 gpurun_out/r04clock gpurun_out/r04c4
tail -15 gpurun_out/r04clock.txt; cat gpurun_out/r04c2/summary.txt; tail -12 gpurun_out/r04c4.txt
