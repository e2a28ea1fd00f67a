#!/bin/bash
# GPU test run as the driver does it, with a per-test time limit.
cd "$(dirname "$0")/.." && exec timeout -k 10 900 python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread "$@"
