#!/bin/bash
# scripts/probes/partition_probe.sh -- can ONE MI355X of this pool show up as several HIP devices?  (VERDICT r5 item 7a:
# the native multi-GPU code has only ever met a one-rank RCCL communicator and the thread-rank stand-in.)
# Reads the compute / memory partitions the box offers and whether this container may change them (the sysfs
# nodes rocm-smi --setcomputepartition writes).  Round 6: SPX / NPS1, DPX / QPX / CPX on offer, but every
# current_compute_partition node is read-only here and the container sees ONE render node of the host's eight
# cards -- no second device can be had on this pool (profiles/r06_partition_probe.txt, EXPERIMENTS.md).
set -u
OUT=gpurun_out/partition
mkdir -p "$OUT"
{
  echo "== devices"; python -c "import torch; print(torch.cuda.device_count())" 2>&1 | tail -1
  echo "== rocm-smi --showcomputepartition"; timeout 60 rocm-smi --showcomputepartition 2>&1 | tail -12
  echo "== rocm-smi --showmemorypartition"; timeout 60 rocm-smi --showmemorypartition 2>&1 | tail -12
  echo "== amd-smi partition"; (timeout 60 amd-smi partition 2>&1 || true) | tail -40
  echo "== sysfs"; for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_compute_partition \
      /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/available_memory_partition; do
      [ -e "$f" ] && echo "$f: $(cat "$f" 2>&1) (writable: $([ -w "$f" ] && echo yes || echo no))"; done
  echo "== nodes"; ls /dev/dri 2>&1 | tr '\n' ' '; echo; ls /sys/class/kfd/kfd/topology/nodes 2>&1 | tr '\n' ' '; echo
} > "$OUT/probe.txt" 2>&1
cat "$OUT/probe.txt"
