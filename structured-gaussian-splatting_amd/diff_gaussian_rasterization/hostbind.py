"""Host placement helper: run the calling process on the CPUs of its GPU's NUMA node (bench.py calls it once per rank; a training script would at start-up)."""
import os

import torch


def bind_to_gpu_numa_node(index: int = 0, all_threads: bool = False):
    """One process per GPU, on the CPUs of the GPU's own NUMA node — what `numactl --cpunodebind` does in a multi-GPU
    launcher.  The step has two host round trips and ~35 launches per 0.8 ms; from the far socket every doorbell and every
    poll of the control block crosses the inter-socket link (measured: the same process runs 0.77-0.79 or 0.84-0.87 ms/step
    depending on where the scheduler happened to put it; 12 interleaved pairs of runs: median 0.788 bound, 0.836 not).
    Binds the calling thread (threads it starts later inherit the mask: the autograd engine's worker); all_threads also
    re-binds the threads that exist already (the HIP runtime's helpers: measured no better).  Returns (node, number of
    CPUs) or None when sysfs does not say."""
    try:
        import glob
        bus = int(torch.cuda.get_device_properties(index).pci_bus_id)
    except Exception:
        return None
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            bdf = os.path.basename(os.readlink(d))                    # 0000:d9:00.0
            if int(bdf.split(":")[1], 16) != bus:
                continue
            node = int(open(os.path.join(d, "numa_node")).read())
            if node < 0:
                return None
            cpus = set()
            for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
                a, _, b = part.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            cpus &= os.sched_getaffinity(0)
            if not cpus:
                return None
            os.sched_setaffinity(0, cpus)
            if all_threads:
                for tid in os.listdir("/proc/self/task"):             # the HIP runtime's helper threads exist already
                    try:
                        os.sched_setaffinity(int(tid), cpus)
                    except OSError:
                        pass
            return node, len(cpus)
        except (OSError, ValueError, IndexError):         # a card without a PCI address / numa_node entry
            continue
    return None
