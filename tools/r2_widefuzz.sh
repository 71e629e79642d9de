#!/bin/bash
# wide fuzz sweep of the front end (byte mutations + tag-list layouts), 40 extra seeds each
out=gpurun_out/r2wf; mkdir -p $out
ZKE_FUZZ_SEEDS=40 timeout -k 10 900 python -m pytest tests/test_taglist.py tests/test_gpu_verify.py -m gpu -q -k "fuzz" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -6 $out/pytest.log
