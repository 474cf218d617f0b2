#!/bin/bash
# ONE entry point for everything that runs ON THE GPU BOX (through gpurun), by sub-command; results go to gpurun_out/<tag>_*.
#   gpurun --timeout 1500 -- 'bash tools/gpu_round.sh tests r04_v1'
#
#   tests   <tag>          pytest -m gpu (whole suite) + smoke()
#   quick   <tag> <expr>   pytest -m gpu -k <expr>
#   bench   <tag> [wl]     the driver's command line (default line incl. `secondary`, cpu_baseline, parity) [+ --workload wl]
#   stages  <tag> [wl]     per-launch HIP-event table of one step (no baselines)
#   ab      <tag> [wl]     same-box A/B: allegro_amd/liballegro_amd_old.so vs the product library, alternated 3 times
#   abn     <tag> <wl> <name>...   same-box A/B/C: the product library vs allegro_amd/liballegro_amd_<name>.so for every name
#   abenv   <tag> <wl> VAR=val ...  same-box A/B/C of environment switches of the Python host (product library; one arm per argument,
#                                   quote several assignments of one arm together)
#   profile <tag> [wl]     rocprofv3 --kernel-trace --stats, then separate --pmc passes, summary + hashed traffic JSON
#   hosts   <tag>          the Python-free hosts (tests/host): C99 driver and C++ AOTInductor package consumer
#   ubench  <tag> <name>   tools/ubench/<name>.bin (hipcc -o it on the build box first: it travels with the snapshot)
#   shards  <tag> [W]      every rank's compact shard of a W-way partition of C4, one after the other (load balance)
#   final   <tag>          tests + bench (c4, c5, c3, c2, c1) + stages + shards + profile c4 / c5: the end-of-round evidence
cd "$(dirname "$0")/.."
CMD=${1:-tests}; TAG=${2:-r06}; ARG=${3:-}
mkdir -p gpurun_out
export TMPDIR=/tmp
case $CMD in
  tests)
    timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/${TAG}_pytest_gpu.log
    timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
    tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -1 gpurun_out/${TAG}_smoke.log ;;
  quick)
    timeout 1500 python -m pytest tests -m gpu -q -k "$ARG" > gpurun_out/${TAG}_pytest_quick.log 2>&1; tail -25 gpurun_out/${TAG}_pytest_quick.log ;;
  bench)
    WL=${ARG:-c4}
    timeout 1200 python bench.py --workload $WL --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_$WL.log 2> gpurun_out/${TAG}_bench_$WL.err
    python tools/bench_brief.py gpurun_out/${TAG}_bench_$WL.log ;;
  stages)
    WL=${ARG:-c4}
    timeout 900 python bench.py --workload $WL --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 \
      > gpurun_out/${TAG}_stagesline_$WL.log 2> gpurun_out/${TAG}_stages_$WL.log
    grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_stagesline_$WL.log | head -1; grep '^\[stage\]' gpurun_out/${TAG}_stages_$WL.log ;;
  ab)
    WL=${ARG:-c4}
    for rep in 1 2 3; do for lib in old new; do
      if [ $lib = old ]; then export ALLEGRO_AMD_LIBRARY=$PWD/allegro_amd/liballegro_amd_old.so; else unset ALLEGRO_AMD_LIBRARY; fi
      r=$(timeout 600 python bench.py --workload $WL --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 2> gpurun_out/${TAG}_ab_$lib.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
      echo "$lib $r | $(grep '^\[stage\]' gpurun_out/${TAG}_ab_$lib.log | awk '{printf "%s %s  ", $2, $3}')"
    done; done | tee gpurun_out/${TAG}_ab_$WL.txt ;;
  abn)
    # same-box A/B/C...: the product library against allegro_amd/liballegro_amd_<name>.so for every further argument, alternated 3 times
    WL=${ARG:-c4}; shift 3
    for rep in 1 2 3; do for lib in new "$@"; do
      if [ $lib = new ]; then unset ALLEGRO_AMD_LIBRARY; else export ALLEGRO_AMD_LIBRARY=$PWD/allegro_amd/liballegro_amd_$lib.so; fi
      r=$(timeout 600 python bench.py --workload $WL --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --sustain 0 2> gpurun_out/${TAG}_abn_$lib.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
      echo "$lib $r | $(grep '^\[stage\]' gpurun_out/${TAG}_abn_$lib.log | awk '{printf "%s %s  ", $2, $3}')"
    done; done | tee gpurun_out/${TAG}_abn_$WL.txt ;;
  abenv)
    # same-box A/B of a host switch: the product library with and without the environment assignment(s) given after the workload
    # (e.g. AA_FUSED_NARROW=1), alternated 3 times
    WL=${ARG:-c4}; shift 3
    for rep in 1 2 3; do for arm in default "$@"; do
      if [ "$arm" = default ]; then pre=""; else pre="$arm"; fi
      r=$(env $pre timeout 600 python bench.py --workload $WL --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --no-secondary --no-md --sustain 0 2> gpurun_out/${TAG}_abenv.log | grep -o '"ms_per_step": [0-9.]*' | head -1)
      echo "[$arm] $r | $(grep '^\[stage\]' gpurun_out/${TAG}_abenv.log | awk '{printf "%s %s  ", $2, $3}')"
    done; done | tee gpurun_out/${TAG}_abenv_$WL.txt ;;
  profile)
    WL=${ARG:-c4}
    bash tools/profile_gpu.sh $WL $TAG > /dev/null 2>&1
    cp gpurun_out/prof_${TAG}_$WL/summary.txt gpurun_out/${TAG}_rocprofv3_${WL}_summary.txt
    python tools/pmc_to_json.py gpurun_out/${TAG}_rocprofv3_${WL}_summary.txt $WL > gpurun_out/${TAG}_pmc_traffic_$WL.json 2> gpurun_out/${TAG}_pmc_to_json.err
    rm -rf gpurun_out/prof_${TAG}_$WL/trace gpurun_out/prof_${TAG}_$WL/pmc_*
    tail -30 gpurun_out/${TAG}_rocprofv3_${WL}_summary.txt ;;
  hosts)
    timeout 1500 python -m pytest tests/test_host_programs.py tests/test_pair_allegro.py -m gpu -q -s > gpurun_out/${TAG}_hosts_full.log 2>&1
    grep -E 'host_|passed|failed|Error|error' gpurun_out/${TAG}_hosts_full.log | cut -c1-400 | tee gpurun_out/${TAG}_hosts.log ;;
  ubench)
    timeout 300 tools/ubench/$ARG.bin > gpurun_out/${TAG}_ubench_$ARG.log 2>&1; cat gpurun_out/${TAG}_ubench_$ARG.log ;;
  shards)
    W=${ARG:-8}
    timeout 900 python bench.py --shard-sweep $W --steps 10 --warmup 3 > gpurun_out/${TAG}_shard_sweep${W}_c4.json 2> /dev/null
    # (RCCL prints its version banner to stdout after the JSON line: keep the first line only)
    head -1 gpurun_out/${TAG}_shard_sweep${W}_c4.json > gpurun_out/.sweep.tmp && mv gpurun_out/.sweep.tmp gpurun_out/${TAG}_shard_sweep${W}_c4.json
    python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_shard_sweep${W}_c4.json')); print({k: d[k] for k in ('max_ms','mean_ms','imbalance_max_over_mean','host_issue_over_gpu_max')})" ;;
  final)
    bash $0 tests $TAG
    bash $0 bench $TAG c4
    for wl in c5 c3 c2 c1; do
      timeout 900 python bench.py --workload $wl --steps $([ $wl = c5 ] && echo 5 || echo 100) --warmup 5 --stages --no-secondary --sustain 0 \
        > gpurun_out/${TAG}_bench_$wl.log 2> gpurun_out/${TAG}_stages_$wl.log
      echo "$wl $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_$wl.log | head -1)"
    done
    bash $0 stages $TAG c4
    bash $0 shards $TAG 8
    bash $0 profile $TAG c4
    bash $0 profile $TAG c5
    for wl in c4 c5; do python tools/pmc_bound_table.py gpurun_out/${TAG}_rocprofv3_${wl}_summary.txt > gpurun_out/${TAG}_bound_table_${wl}.md; done
    timeout 600 python tools/md_loop.py --workload c4 --steps 100 > gpurun_out/${TAG}_md_loop_c4.json 2> /dev/null
    timeout 600 python tools/md_loop.py --workload c4 --steps 200 --dt 0.5 > gpurun_out/${TAG}_md_loop_c4_dt0.5.json 2> /dev/null
    timeout 600 python tools/md_loop.py --workload c3 --steps 100 > gpurun_out/${TAG}_md_loop_c3.json 2> /dev/null
    timeout 600 python bench.py --mode train-step --workload c3 --steps 5 --warmup 2 > gpurun_out/${TAG}_train_step_c3.json 2> /dev/null
    AA_TRAIN_EAGER=1 timeout 600 python bench.py --mode train-step --workload c3 --steps 5 --warmup 2 --no-gpu-reference > gpurun_out/${TAG}_train_step_c3_eager.json 2> /dev/null
    timeout 900 python bench.py --mode train-step --workload c4 --steps 2 --warmup 1 --no-gpu-reference --train-chunk-edges 400000 > gpurun_out/${TAG}_train_step_c4_chunked.json 2> /dev/null
    grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_train_step_c3.json gpurun_out/${TAG}_train_step_c3_eager.json gpurun_out/${TAG}_train_step_c4_chunked.json
    timeout 400 python tools/train_profile.py > gpurun_out/${TAG}_train_profile_c3.txt 2> /dev/null
    timeout 400 python tools/train_profile.py --stacks aten::copy_,aten::add_,aten::add,aten::cat,aten::fill_,aten::mul --top 40 > gpurun_out/${TAG}_train_nodes_c3.txt 2> /dev/null
    # the driver's N > 1 launch line on this one-GPU box: N ranks share cuda:0, rows staged through the host (gloo) -- exercises bench.py's
    # sharded branch end to end (rendezvous, slab shards, both exchanges, barrier + max-over-ranks clock); its ms/step is N shards on ONE GPU
    for N in 2 8; do
      # (round 6: bench.py launches itself under torch.distributed.run when WORLD_SIZE is unset -- the driver's own command shape)
      AA_BENCH_BACKEND=gloo AA_BENCH_DEVICE=0 timeout 600 python bench.py --gpus $N --steps 5 --warmup 2 --sustain 0 2> /dev/null | grep '^{' | head -1 > gpurun_out/${TAG}_bench_launch_line_x$N.json
      echo "launch line N=$N $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_launch_line_x$N.json | head -1)"
    done ;;
  *) echo "unknown sub-command $CMD"; exit 2 ;;
esac
