#!/bin/bash
# The closing GPU session of a round.  Round 6: tools/gpu_r6z.sh -- PMC traffic first (so that the bench line of the same
# session finds a record stamped with this build's kernel-source hash; tests/test_host_logic.py checks the committed pair),
# the whole GPU suite, the default bench line, rocprofv3 kernel stats of the full-length command (exact and emulated), call
# sequences, training step, micro-benchmarks, the DSBDD_EMU=6 gate run of the suite.  Usage: tools/gpu_final.sh TAG
exec bash "$(dirname "$0")/gpu_r6z.sh" "${1:-final}"
