#!/bin/bash
# whole GPU suite, smoke, both bench lines, PMC instruction counts
out=gpurun_out/r2a; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
B="--no-cpu --no-saturated"
python bench.py $B --steps 20 --warmup 5 > $out/bench20.json 2>$out/bench20.err || { tail -5 $out/bench20.err; exit 1; }
python -c "import json; j=json.load(open('$out/bench20.json')); print('burst', j['value'], j['ms_per_step'])"
python bench.py $B --steps 2000 --warmup 100 > $out/bench2000.json 2>$out/bench2000.err || { tail -5 $out/bench2000.err; exit 1; }
python -c "import json; j=json.load(open('$out/bench2000.json')); print('steady', j['value'], j['ms_per_step'], j['kernels_us_alone'])"
rm -rf $out/p_instr
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $out/p_instr -o runc --output-format csv -- python bench.py $B --steps 8 --warmup 4 --streams 1 --alone-steps 0 > $out/p_instr.log 2>&1 || { tail -5 $out/p_instr.log; exit 1; }
python tools/pmc_summary.py $out/p_instr > $out/c2_instr_pmc.json; python -c "
import json; j=json.load(open('$out/c2_instr_pmc.json'))
for k,v in j.items(): print(k, {a: round(b/1024) for a,b in v.items() if a.startswith('SQ_INSTS') or a=='SQ_WAVE_CYCLES'})"
