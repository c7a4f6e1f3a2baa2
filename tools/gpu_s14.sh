#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "moe_ or fused_ or decoder_harness" > $O/s14_pytest.log 2>&1
tail -30 $O/s14_pytest.log
