#!/bin/bash
O=gpurun_out/r02; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "skinny or bf16" 2>&1 | tail -8
