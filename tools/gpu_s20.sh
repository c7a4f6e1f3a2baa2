#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/s20_pytest.log 2>&1
tail -6 $O/s20_pytest.log
