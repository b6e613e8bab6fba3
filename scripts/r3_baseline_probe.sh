#!/bin/bash
# Round-3 starting measurements of the round-2 kernels (what the work of this round is planned against):
# cfg 5 kernel stats + SQ counters, the default-solver kernel's SQ counters, launch gaps at 1448² and 256².
set -o pipefail
O=gpurun_out/r3a
mkdir -p $O
export TMPDIR=/tmp
python scripts/cfg5_probe.py > $O/cfg5_probe.json 2> $O/cfg5_probe.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg5_stats -- python3 scripts/cfg5_probe.py > $O/cfg5_stats.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/cfg5_pmc_sq -- python3 scripts/cfg5_probe.py > $O/cfg5_pmc_sq.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/auto_gen_pmc_sq -- python3 bench.py --steps 6 --warmup 5 --no-cpu --no-secondary --solver AutoTsit5 --winds 10,3 > $O/auto_gen_pmc_sq.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/auto_sym_pmc_sq -- python3 bench.py --steps 6 --warmup 5 --no-cpu --no-secondary --solver AutoTsit5 > $O/auto_sym_pmc_sq.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/gap_1448 -- python3 bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --no-events --grid-n 1448 > $O/gap_1448.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/gap_256 -- python3 bench.py --steps 200 --warmup 5 --no-cpu --no-secondary --no-events --grid-n 256 > $O/gap_256.log 2>&1 &&
rocprofv3 --kernel-trace --output-format csv -d $O/gap_4096 -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --no-events > $O/gap_4096.log 2>&1 &&
python bench.py --steps 200 --warmup 5 --no-cpu --no-secondary --grid-n 256 > $O/bench_256.json 2> $O/bench_256.err &&
python bench.py --steps 200 --warmup 5 --no-cpu --no-secondary --grid-n 256 --no-events > $O/bench_256_noev.json 2>> $O/bench_256.err &&
python bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --grid-n 1448 --no-events > $O/bench_1448_noev.json 2>> $O/bench_256.err &&
echo "collected r3a"
