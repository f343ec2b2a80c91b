#!/bin/bash
ALGP_GEMM_VARIANT=5 timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
