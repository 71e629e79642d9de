#!/bin/bash
out=gpurun_out/r2j; mkdir -p $out
export TMPDIR=/tmp
timeout -k 5 120 ./tools/ubench/launch_rate 2000 > $out/launch_rate.txt 2>&1 || { cat $out/launch_rate.txt; exit 1; }
cat $out/launch_rate.txt
rm -rf $out/p_instr
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/p_instr -o runc --output-format csv -- python bench.py --steps 6 --warmup 4 --no-cpu --no-saturated --streams 1 --alone-steps 0 > $out/p_instr.log 2>&1 || { tail -5 $out/p_instr.log; exit 1; }
python tools/pmc_summary.py $out/p_instr > $out/c2_instr_pmc.json; cat $out/c2_instr_pmc.json
