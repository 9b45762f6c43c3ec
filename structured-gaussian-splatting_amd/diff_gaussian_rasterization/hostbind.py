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
        props = torch.cuda.get_device_properties(index)
        want = (int(getattr(props, "pci_domain_id", 0)), int(props.pci_bus_id), int(getattr(props, "pci_device_id", 0)))
    except Exception:
        return None
    # match domain, bus and device together: on multi-domain hosts two cards can share a bus number, and the wrong socket's CPUs
    # are exactly what this helper exists to avoid; more than one match -> do nothing
    found = []
    for d in glob.glob("/sys/class/drm/card*/device"):
        try:
            bdf = os.path.basename(os.readlink(d))                    # 0000:d9:00.0
            dom, bus, rest = bdf.split(":")
            if (int(dom, 16), int(bus, 16), int(rest.split(".")[0], 16)) == want:
                found.append(d)
        except (OSError, ValueError, IndexError):         # a card without a PCI address
            continue
    if len({os.path.realpath(d) for d in found}) != 1:
        return None
    for d in found[:1]:
        try:
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
