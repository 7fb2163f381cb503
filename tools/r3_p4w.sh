#!/bin/bash
# Runs ON THE GPU BOX: the stand-alone probe of the 4-wave / 128x128-per-wave GEMM core (tools/p4w_probe.hip), every prebuilt variant
O=gpurun_out/p4w; mkdir -p $O
for b in tools/p4w_probe_il*; do echo "== $b"; timeout 120 $b 2>&1 | tail -12; done | tee $O/p4w.log
