#!/bin/bash
# Round-3 counter passes (profiles/run_pmc.sh) for the workloads of DESIGN.md section 5: usage: bash profiles/r03_pmc_all.sh <tag> <batch 1|2>
set -u
TAG=$1; B=$2
P="bash profiles/run_pmc.sh $TAG"
if [ "$B" = 1 ]; then
  $P config0_beam --config 0 --no-end-to-end && $P config0_naive --config 0 --mode naive --no-end-to-end && \
  $P config0_beam_fill --config 0 --camera fill --no-end-to-end && $P config0_naive_fill --config 0 --camera fill --mode naive --no-end-to-end && \
  $P config0_beam_inside --config 0 --camera inside --no-end-to-end
else
  $P config0_beam_startup --config 0 --camera startup --no-end-to-end && $P config2_beam --config 2 --no-end-to-end && \
  $P config3_beam --config 3 --no-end-to-end && $P config4_beam --config 4 --no-end-to-end && $P config5_beam --config 5 --no-end-to-end
fi
