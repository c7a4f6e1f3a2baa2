#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fused_ or attn_decode or decoder_harness" > $O/s6_pytest.log 2>&1
tail -60 $O/s6_pytest.log
