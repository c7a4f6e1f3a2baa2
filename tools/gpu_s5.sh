#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "pack_library or prepacked or hf_from or tp_sharded or ksplit_grid or workspace" > $O/s5_pytest.log 2>&1
tail -40 $O/s5_pytest.log
