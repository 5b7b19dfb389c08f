#!/bin/bash
# One gpurun call: parity tests, smoke, a bench line and (optionally) a rocprofv3 kernel-trace of the bench.
# usage: scripts/gpu_round.sh [tests|bench|prof|all]  (everything lands in gpurun_out/)
set -u
mode=${1:-all}
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd "${GRAFT_REPO_ROOT:-.}"
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt
lscpu | grep -E "Model name|^CPU\(s\)" >> gpurun_out/rocminfo.txt
if [[ $mode == tests || $mode == all ]]; then
  rm -f gpurun_out/kernel_report.jsonl
  : > gpurun_out/pytest.log
  if [[ -n "${PYTEST_SPLIT:-}" ]]; then
    # one process per test function: a memory fault in one kernel does not hide the verdicts of the others
    for t in $(python -m pytest tests -m gpu --collect-only -q 2>/dev/null | grep "::" | sed 's/\[.*//' | sort -u); do
      echo "=== $t" >> gpurun_out/pytest.log
      timeout 600 python -m pytest "$t" -m gpu -q --timeout=300 2>&1 | tail -25 >> gpurun_out/pytest.log
    done
  else
    timeout 1500 python -m pytest tests -m gpu -q --timeout=600 ${PYTEST_ARGS:-} >> gpurun_out/pytest.log 2>&1
    echo "pytest exit $?" >> gpurun_out/pytest.log
  fi
  grep -E "^===|passed|failed|error|Error|assert|xfail" gpurun_out/pytest.log | tail -60
  timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [[ $mode == bench || $mode == all ]]; then
  timeout 900 python bench.py ${BENCH_ARGS:---steps 2 --warmup 1} > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
  tail -5 gpurun_out/bench.log
fi
if [[ $mode == prof || $mode == all ]]; then
  export TMPDIR=/tmp
  out=$PWD/gpurun_out/prof
  rm -rf $out; mkdir -p $out
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $OLDPWD/bench.py ${PROF_ARGS:---steps 1 --warmup 1 --no-cpu-baseline} > $out/run.log 2>&1; echo "prof exit $?" >> $out/run.log)
  if [[ -n "${PMC:-}" ]]; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
      (cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_$ctr -o b -- python $OLDPWD/bench.py --steps 1 --warmup 0 --global-batch ${PMC_BATCH:-1024} --no-cpu-baseline > $out/pmc_$ctr.log 2>&1)
      python $OLDPWD/scripts/pmc_summary.py $out/pmc_$ctr/b_counter_collection.csv $ctr > $out/pmc_${ctr}_summary.txt 2>&1
      rm -rf $out/pmc_$ctr
    done
    python $OLDPWD/scripts/pmc_traffic.py $out/pmc_FETCH_SIZE_summary.txt $out/pmc_WRITE_SIZE_summary.txt ${PMC_BATCH:-1024} > $out/pmc_gemm_traffic.json 2>&1
  fi
  tail -3 $out/run.log
  find $out -name "*kernel_stats*" | head
  f=$(find $out -name "*kernel_stats*.csv" | head -1)
  [[ -n "$f" ]] && head -25 "$f"
  t=$(find $out -name "*kernel_trace*.csv" | head -1)
  [[ -n "$t" ]] && python $OLDPWD/scripts/prof_summary.py "$t" > $out/kernel_summary.txt 2>&1
  # keep the merge-back small: drop the raw per-dispatch trace if it is large
  find $out -name "*kernel_trace*.csv" -size +20M -delete
fi
