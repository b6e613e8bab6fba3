#!/bin/bash
# Round measurement suite, run on the GPU box from the repo root:
#   gpurun --timeout 1100 -- 'bash scripts/collect_profiles.sh r2'
# then, back in the build container:  python scripts/make_profiles.py r2
# Each rocprofv3 pass is its own process; the --pmc passes carry no trace domains.
set -o pipefail
R=${1:-r4}
O=gpurun_out
rm -rf $O/${R}_stats $O/${R}_stats_generic $O/${R}_stats_ring $O/${R}_pmc_sq $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_sq_auto $O/${R}_pmc_sq_auto_generic $O/${R}_cfg5_stats $O/${R}_cfg5_pmc_sq $O/${R}_cfg5_pmc_fetch $O/${R}_cfg5_pmc_write
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/${R}_bench.json 2> $O/${R}_bench.err &&
python bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --winds 10,3 > $O/${R}_bench_generic.json 2>> $O/${R}_bench.err &&
python bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --ring-of-one --grid-n 1448 > $O/${R}_bench_ring_of_one_1448.json 2>> $O/${R}_bench.err &&
python bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --grid-n 1448 > $O/${R}_bench_1448.json 2>> $O/${R}_bench.err &&
python bench.py --steps 20 --warmup 5 --no-cpu --no-secondary --ring-of-one > $O/${R}_bench_ring_of_one_4096.json 2>> $O/${R}_bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu --no-secondary > $O/${R}_stats.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats_generic -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-secondary --winds 10,3 > $O/${R}_stats_generic.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_stats_ring -- python3 bench.py --steps 50 --warmup 5 --no-cpu --no-secondary --ring-of-one --grid-n 1448 > $O/${R}_stats_ring.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/${R}_pmc_sq -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/${R}_pmc_sq.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/${R}_pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu --no-secondary > $O/${R}_pmc_write.log 2>&1 &&
PMCSQ="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
rocprofv3 --pmc $PMCSQ --output-format csv -d $O/${R}_pmc_sq_auto -- python3 bench.py --steps 6 --warmup 5 --no-cpu --no-secondary --solver AutoTsit5 > $O/${R}_pmc_sq_auto.log 2>&1 &&
rocprofv3 --pmc $PMCSQ --output-format csv -d $O/${R}_pmc_sq_auto_generic -- python3 bench.py --steps 6 --warmup 5 --no-cpu --no-secondary --solver AutoTsit5 --winds 10,3 > $O/${R}_pmc_sq_auto_generic.log 2>&1 &&
python scripts/cfg5_profile.py 58 > $O/${R}_cfg5_profile.jsonl 2> $O/${R}_cfg5_profile.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${R}_cfg5_stats -- python3 scripts/cfg5_profile.py 58 > $O/${R}_cfg5_stats.log 2>&1 &&
rocprofv3 --pmc $PMCSQ --output-format csv -d $O/${R}_cfg5_pmc_sq -- python3 scripts/cfg5_profile.py 20 > $O/${R}_cfg5_pmc_sq.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${R}_cfg5_pmc_fetch -- python3 scripts/cfg5_profile.py 20 > $O/${R}_cfg5_pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${R}_cfg5_pmc_write -- python3 scripts/cfg5_profile.py 20 > $O/${R}_cfg5_pmc_write.log 2>&1 &&
python scripts/baseline_configs.py 2> /dev/null > $O/${R}_baseline_configs.jsonl &&
echo "collected $R"
