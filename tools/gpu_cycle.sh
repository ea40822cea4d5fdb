#!/bin/bash
# GPU-box cycle.  Usage (repo root, via gpurun): bash tools/gpu_cycle.sh <tag> [tests|bench|benchq|reh|trace|pmc|pmcw|melpmc|beats|beatspmc|beatswait|surf|host|second|smoke|ab:<libA>:<libB> ...]
tag=${1:-x}; shift
what=${@:-tests bench}
export TMPDIR=/tmp
mkdir -p gpurun_out
# rocprofv3 databases are summarised on the box and then deleted: gpurun merges at most 64 MiB of gpurun_out/ back
for w in $what; do
case $w in
tests) timeout 2400 python -m pytest tests -m gpu -q -s --durations=8 2>&1 | grep -v "^$" | tail -40 | cut -c1-400 | tee gpurun_out/tests_$tag.log ;;
bench) timeout 600 python bench.py 2>gpurun_out/bench_$tag.err | tail -1 > gpurun_out/bench_$tag.json; cut -c1-600 gpurun_out/bench_$tag.json; tail -3 gpurun_out/bench_$tag.err ;;
benchq) timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/benchq_$tag.err | tail -1 > gpurun_out/benchq_$tag.json; cut -c1-300 gpurun_out/benchq_$tag.json; tail -3 gpurun_out/benchq_$tag.err ;;
reh) timeout 300 env NCCL_DEBUG=INFO python bench.py --steps 50 --warmup 10 --no-cpu-baseline --rehearse-exchange 2>gpurun_out/reh_$tag.err | tail -1 > gpurun_out/reh_$tag.json
     python -c "import json;d=json.load(open('gpurun_out/reh_$tag.json'));print('rehearsal', d['ms_per_step']);print(json.dumps(d['dist']['exchange_tail_us'],indent=0));print(d['dist']['rccl_debug'])" ;;
trace) timeout 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o $tag -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline $BENCH_ARGS > gpurun_out/prof_$tag.log 2>&1
       python tools/rocpd_summary.py gpurun_out/prof_$tag/${tag}_results.db 0.75 > gpurun_out/prof_$tag.md 2>&1; head -60 gpurun_out/prof_$tag.md | cut -c1-120
       python tools/step_timeline.py gpurun_out/prof_$tag/${tag}_results.db 0 0 > gpurun_out/timeline_$tag.md 2>&1; tail -1 gpurun_out/timeline_$tag.md
       rm -rf gpurun_out/prof_$tag ;;
pmc) for pass in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
       set -- $pass; name=$1; shift
       timeout 400 rocprofv3 --kernel-trace --pmc $@ -d gpurun_out/pmc_${tag}_$name -o $name -- python bench.py --surface driver --steps 2 --warmup 1 --no-graph --no-cpu-baseline > gpurun_out/pmc_${tag}_$name.log 2>&1
     done
     python tools/pmc_ratio_summary.py gpurun_out/pmc_${tag}_mfma/mfma_results.db > gpurun_out/pmc_${tag}_mfma.md 2>&1
     python tools/pmc_summary.py gpurun_out/pmc_${tag}_fetch/fetch_results.db > gpurun_out/pmc_${tag}_fetch.md 2>&1
     python tools/pmc_summary.py gpurun_out/pmc_${tag}_write/write_results.db > gpurun_out/pmc_${tag}_write.md 2>&1
     python tools/pmc_traffic_json.py gpurun_out/pmc_${tag}_fetch/fetch_results.db gpurun_out/pmc_${tag}_write/write_results.db 9 > gpurun_out/pmc_${tag}_traffic.json 2>gpurun_out/pmc_${tag}_traffic.err
     head -30 gpurun_out/pmc_${tag}_mfma.md | cut -c1-200
     rm -rf gpurun_out/pmc_${tag}_mfma gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write ;;
pmcw) bash tools/pmc_wait.sh $tag > /dev/null 2>&1; grep -E "block0|conv0_kernel|glu128|kernel \|" gpurun_out/pmcw_$tag.md | cut -c1-110
      rm -rf gpurun_out/pmcw_${tag}_a gpurun_out/pmcw_${tag}_b gpurun_out/pmcw_${tag}_c ;;
lt) timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "lightning" 2>&1 | grep -v "^$" | tail -30 | cut -c1-400 | tee gpurun_out/tests_lt_$tag.log ;;
surf) # the four launch paths of the same step, same box: Lightning-order whole-step (headline), driver by hand, and the two eager forms
      for v in "lightning" "driver" "lightning --no-graph" "driver --no-graph --prefetch off"; do
        n=$(echo $v | tr -d ' -'); timeout 300 python bench.py --surface $v --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/surf_${tag}_$n.err | tail -1 > gpurun_out/surf_${tag}_$n.json
        python -c "import json;d=json.load(open('gpurun_out/surf_${tag}_$n.json'));print('$v', d['ms_per_step'], d['config']['surface'].get('driver_surface_ms_per_step'), d['config']['launch'][:40])" 2>&1 | tail -1
      done ;;
tsprobe) timeout 300 python bench.py --ts-probe --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/tsprobe_$tag.err | tail -1 > gpurun_out/tsprobe_$tag.json
         python -c "import json;d=json.load(open('gpurun_out/tsprobe_$tag.json'));print('ts-probe run', d['ms_per_step']);[print('  %-40s %8.1f us' % (t, u)) for t, u in d.get('ts_probe_us', [])]" ;;
beats) timeout 300 python tools/beats_bench.py 2>/dev/null | tail -1 > gpurun_out/beats_$tag.json; cut -c1-400 gpurun_out/beats_$tag.json ;;
host) timeout 300 python bench.py --host-batches --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/host_$tag.json; python -c "import json;d=json.load(open('gpurun_out/host_$tag.json'));print('host batches (PCIe-inclusive)', d['ms_per_step'], d['value'])" ;;
beatspmc) timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmcbeats_$tag -o b -- python tools/beats_bench.py > gpurun_out/pmcbeats_$tag.log 2>&1
          python tools/pmc_ratio_summary.py gpurun_out/pmcbeats_$tag/b_results.db > gpurun_out/pmcbeats_$tag.md 2>&1; head -12 gpurun_out/pmcbeats_$tag.md | cut -c1-200; rm -rf gpurun_out/pmcbeats_$tag ;;
beatswait) bash tools/pmc_beats.sh $tag 2>&1 | tail -8 | cut -c1-300 ;;
beatsfetch) for c in FETCH_SIZE WRITE_SIZE; do timeout 300 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmcbf_${tag}_$c -o b -- python tools/beats_bench.py > gpurun_out/pmcbf_${tag}_$c.log 2>&1; done
            python tools/pmc_summary.py gpurun_out/pmcbf_${tag}_FETCH_SIZE/b_results.db > gpurun_out/pmcbeats_fetch_$tag.md 2>&1; python tools/pmc_summary.py gpurun_out/pmcbf_${tag}_WRITE_SIZE/b_results.db > gpurun_out/pmcbeats_write_$tag.md 2>&1
            head -12 gpurun_out/pmcbeats_fetch_$tag.md | cut -c1-200; rm -rf gpurun_out/pmcbf_${tag}_FETCH_SIZE gpurun_out/pmcbf_${tag}_WRITE_SIZE ;;
melpmc) bash tools/pmc_mel.sh $tag 2>&1 | tail -3 | cut -c1-300; rm -rf gpurun_out/pmcmel_${tag}_[abcfw] ;;
smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
second) timeout 300 python tools/bench_2024.py --graph --prefetch 2>/dev/null | tail -1 > gpurun_out/bench2024_$tag.json; cut -c1-80,330- gpurun_out/bench2024_$tag.json
        timeout 300 python tools/bench_2024.py --graph 2>/dev/null | tail -1 > gpurun_out/bench2024_inline_$tag.json
        timeout 300 python bench.py --embeddings --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_emb_$tag.json; cut -c1-300 gpurun_out/bench_emb_$tag.json ;;
ab:*) # same-box A/B of two builds of the library (tools/build_variant.py / a saved copy): ab:<libA>:<libB>, alternated twice
     IFS=: read -r _ la lb <<< "$w"
     for rep in 1 2; do for lib in $la $lb; do
       n=$(basename $lib .so)
       timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --lib $lib --dump-launches gpurun_out/ab_${tag}_${n}_$rep.launches.json 2>gpurun_out/ab_${tag}_${n}_$rep.err | tail -1 > gpurun_out/ab_${tag}_${n}_$rep.json
       python -c "import json;d=json.load(open('gpurun_out/ab_${tag}_${n}_$rep.json'));print('$n', $rep, d['ms_per_step'])" 2>&1 | tail -1
     done; done ;;
esac
done
