#!/bin/bash
# Runs GPU steps in order on the gpurun box; a step that times out / is killed stops the chain
# (no further GPU work after a hang).  Usage: tools/gpu_run.sh step1 step2 ...
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
run() {  # name, timeout, command...
    local name=$1 t=$2; shift 2
    echo "== $name" | tee -a "$OUT/steps.log"
    timeout -k 10 "$t" "$@"
    local rc=$?
    echo "== $name exit=$rc" | tee -a "$OUT/steps.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name was killed: stopping"; exit 1; fi
    return 0
}
for step in "$@"; do
case $step in
tests) run tests 900 bash -c "python -m pytest tests -m gpu -q -x --timeout 600 > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log" ;;
testsall) run tests 900 bash -c "python -m pytest tests -m gpu -q --timeout 600 > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log" ;;
smoke) run smoke 300 bash -c "python -c 'import __graft_entry__ as g; g.smoke()' > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log" ;;
bench) run bench 420 bash -c "python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err" ;;
prof) run prof 420 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err; tail -3 $OUT/prof.err; find $OUT/prof -name '*stats*' | head" ;;
pmc) run pmc 420 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --batch 256 --pool 1024 --no-cpu-baseline --no-secondary --no-self-check > $OUT/pmc1.json 2> $OUT/pmc1.err; tail -2 $OUT/pmc1.err"
     run pmc2 420 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --batch 256 --pool 1024 --no-cpu-baseline --no-secondary --no-self-check > $OUT/pmc2.json 2> $OUT/pmc2.err; tail -2 $OUT/pmc2.err" ;;
pmcsq) B="--steps 3 --warmup 1 --batch 256 --pool 1024 --no-cpu-baseline --no-secondary --no-self-check --no-overlap-match"
     run pmcsq1 420 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq1 -- python3 $ROOT/bench.py $B > $OUT/pmcsq1.json 2> $OUT/pmcsq1.err; tail -2 $OUT/pmcsq1.err"
     run pmcsq2 420 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq2 -- python3 $ROOT/bench.py $B > $OUT/pmcsq2.json 2> $OUT/pmcsq2.err; tail -2 $OUT/pmcsq2.err" ;;
diag) run diag 420 bash -c "python tools/gpu_diag.py > $OUT/diag.log 2>&1; tail -20 $OUT/diag.log" ;;
profserial) run profserial 420 bash -c "cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 $ROOT/bench.py --steps 10 --warmup 2 --batch 256 --no-cpu-baseline --no-secondary --no-self-check --no-overlap-match > $OUT/prof_serial.json 2> $OUT/prof_serial.err; tail -3 $OUT/prof_serial.err" ;;
extra) run extra 420 bash -c "python tools/bench_extra.py c1 c3 c4 pcie > $OUT/extra.json 2> $OUT/extra.err; cat $OUT/extra.json; tail -5 $OUT/extra.err" ;;
ubench) run ubench 300 bash -c "cd tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o valu_rate valu_rate.hip && ./valu_rate > $OUT/valu_rate.txt 2>&1; cat $OUT/valu_rate.txt" ;;
ab) # A/B of schedule options on the C2 bench line (40 steps each, same process settings otherwise)
    for v in ${AB_VARIANTS:-"best:" "overlap:--schedule@overlap" "p4:--parts@4" "p2:--parts@2" "serial:--schedule@serial" "best2:"}; do
        name=${v%%:*}; rest=${v#*:}; envs=""; flags=""
        for t in $rest; do case $t in @*) envs="$envs ${t#@}";; *) flags="$flags ${t//@/ }";; esac; done
        run ab_$name 300 bash -c "env $envs python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary $flags > $OUT/ab_$name.json 2> $OUT/ab_$name.err; python -c \"import json; j=json.loads(open('$OUT/ab_$name.json').read().strip().splitlines()[-1]); print('$name', round(j['value']), round(j['ms_per_step'],4), j['timed_region_repeats']['frames_per_s'], {k: v['ms'] for k, v in j['stages'].items()})\""
    done ;;
earlyout) run earlyout 300 bash -c "python tools/early_out_ab.py > $OUT/early_out.json 2> $OUT/early_out.err; cat $OUT/early_out.json; tail -3 $OUT/early_out.err" ;;
exttests) run exttests 900 bash -c "python -m pytest tests/test_gpu_extractor.py tests/test_gpu_threads.py tests/test_gpu_pipeline.py -m gpu -q -x --timeout 600 > $OUT/pytest_ext.log 2>&1; tail -15 $OUT/pytest_ext.log" ;;
*) echo "unknown step $step" ;;
esac
done
