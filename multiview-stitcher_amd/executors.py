"""Multi-GPU farms behind the reference's own seams: independent pairs and output chunks,
no collectives.

``DevicePairExecutor``  -> ``registration.register(..., pairwise_executor=...)``   (registration.py:2634-2655)
``fuse_on_devices``     -> the role of ``batch_options["batch_func"]``            (fusion/_core.py:1133-1141)
``process_batch_using_gpus`` -> a ``batch_options["batch_func"]`` like the reference's process_batch_using_joblib / _ray /
                           _dask (misc_utils.py:161-234): the blocks of a batch are fused on several GPUs of the node
``shard_units``         -> rank-local work list for one-process-per-GPU launches (torch.distributed ranks)
"""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np


def shard_units(n_units, world_size, rank, weights=None):
    """Indices of the units rank ``rank`` owns.  Without weights: contiguous blocks (keeps neighbouring
    chunks, hence shared tiles, on one GPU).  With weights (e.g. overlap voxels): greedy longest-first."""
    if weights is None:
        bounds = np.linspace(0, n_units, world_size + 1).astype(int)
        return list(range(bounds[rank], bounds[rank + 1]))
    order = np.argsort(-np.asarray(weights, dtype=float), kind="stable")
    loads = np.zeros(world_size)
    owner = np.empty(n_units, dtype=int)
    for u in order:
        r = int(np.argmin(loads))
        owner[u] = r
        loads[r] += weights[u]
    return [int(u) for u in range(n_units) if owner[u] == rank]


class DevicePairExecutor:
    """pairwise_executor: registers edge k on device ``devices[k % len(devices)]`` from one host thread per
    device (the library serialises per device and releases the GIL inside calls)."""

    def __init__(self, devices=(0,)):
        self.devices = list(devices)

    def __call__(self, msims, edges, register_kwargs):
        from . import registration
        from .device import DeviceArray, is_device_array

        n = len(self.devices)
        buckets = [[k for k in range(len(edges)) if k % n == d] for d in range(n)]
        results = [None] * len(edges)

        def work(d):
            dev = self.devices[d]
            cache = registration._BinCache()
            local_views = {}        # view index -> the view as this device sees it (one peer copy per tile and device)

            def local(iv):
                if iv not in local_views:
                    m = msims[iv]
                    data = m.data if not hasattr(m, "scales") else None
                    if data is not None and is_device_array(data) and (data.device & 0xff) != (dev & 0xff):   # another GPU (the high bits are a context lane)
                        m = m.copy(data=data.on_device(dev))   # peer copy over xGMI
                    local_views[iv] = m
                return local_views[iv]

            for k in buckets[d]:
                i, j = edges[k]
                results[k] = registration.register_pair_of_msims(local(i), local(j), device=dev, _bin_cache=cache, **register_kwargs)

        with ThreadPoolExecutor(max_workers=n) as ex:
            list(ex.map(work, range(n)))
        return results


def process_batch_using_gpus(func, block_ids, devices=(0,)):
    """batch_func for ``fusion.fuse(..., output_zarr_url=..., batch_options={"batch_func": process_batch_using_gpus,
    "n_batch": k, "batch_func_kwargs": {"devices": (0, 1, ...)}})`` -- the GPU counterpart of the reference's
    process_batch_using_joblib (misc_utils.py:184-209): ``func`` is fuse()'s ``fuse_chunk(block_id, device=...)``; the
    blocks of the batch are dealt to the devices in contiguous runs (neighbouring blocks share tiles, so a device's
    peer-copied tiles are reused) and fused from one host thread per device (the library releases the GIL inside calls;
    every block writes its own region of the output store, so there is nothing to merge)."""
    devices = list(devices)
    n = len(devices)
    block_ids = list(block_ids)
    bounds = np.linspace(0, len(block_ids), n + 1).astype(int)

    def work(d):
        for b in block_ids[bounds[d]:bounds[d + 1]]:
            func(b, device=devices[d])

    if n == 1:
        work(0)
        return
    with ThreadPoolExecutor(max_workers=n) as ex:
        list(ex.map(work, range(n)))


def fuse_on_devices(sims, devices=(0,), **fuse_kwargs):
    """Fuse with the output chunks farmed over several GPUs of one process: chunk b goes to device
    ``devices[b % n]`` (z-major block order keeps a device's chunks adjacent); the host array is assembled
    from the per-device partial results -- or, with ``output_zarr_url``, every device streams its chunks into the one
    Zarr store and there is nothing to assemble.  Equivalent to fusion.fuse on one device."""
    from . import fusion

    n = len(devices)
    counter = {}

    def make_filter(d):
        def f(block_index):
            key = tuple(block_index)
            if key not in counter:
                counter[key] = len(counter)
            return counter[key] % n == d
        return f

    zarr_url = fuse_kwargs.get("output_zarr_url")
    if zarr_url is not None:
        import os
        import shutil

        if (fuse_kwargs.get("zarr_options") or {}).get("overwrite", True) and os.path.exists(zarr_url):
            shutil.rmtree(zarr_url)
    # enumerate blocks deterministically first (single pass with a rejecting filter; creates the output store, if any)
    fusion.fuse(sims, chunk_filter=lambda bi: counter.setdefault(tuple(bi), len(counter)) < 0, device=devices[0], **fuse_kwargs)
    parts = [None] * n

    def work(d):
        parts[d] = fusion.fuse(sims, chunk_filter=lambda bi: counter[tuple(bi)] % n == d, device=devices[d], **fuse_kwargs)

    with ThreadPoolExecutor(max_workers=n) as ex:
        list(ex.map(work, range(n)))
    out = parts[0]
    if zarr_url is not None:
        # every worker wrote its own chunk files into the one store: nothing to merge; the pyramid is built once
        zopt = fuse_kwargs.get("zarr_options") or {}
        if zopt.get("ome_zarr", False):
            from . import ngff_utils

            out = ngff_utils.write_sim_to_ome_zarr(out, zarr_url, overwrite=False, ngff_version=zopt.get("ngff_version", "0.4"),
                                                   zarr_array_creation_kwargs=zopt.get("zarr_array_creation_kwargs"), device=devices[0])
        return out
    data = np.asarray(out.data).copy()
    for p in parts[1:]:
        data += np.asarray(p.data)   # disjoint chunks, untouched ones are zero
    out.data = data
    return out


def cpu_quota_cores():
    """CPU quota of this container in cores (cgroup v2 ``cpu.max`` / v1 cfs quota); None when unlimited or unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            p = float(f.read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def _default_block():
    """CPUs per process block: 16 (measured: 16 and 32 CPUs are equally good, 8 cost 2 ms of a 62-ms step), fewer only when the
    container's quota is smaller -- a larger quota does not widen the block (the point is compactness; the ranks of a node take
    consecutive blocks)."""
    quota = cpu_quota_cores()
    return 16 if (quota is None or quota >= 16) else max(8, int(round(quota)))


def pin_process_to_compact_cpus(slot=0, n_cpus=None):
    """Restrict this process (and every thread it starts afterwards: the pair workers of ``compute_pairwise_registrations``,
    the HIP runtime's helper threads) to one compact block of CPUs.  Call it BEFORE the first HIP / torch call.

    Why: the pairwise registrations are driven by 16 host threads that wake up for tens of microseconds between library
    calls.  On a two-socket host with 256 hardware threads and no affinity the scheduler spreads and migrates them over
    both sockets (cold caches, cross-socket wake-ups, and a cgroup quota that is charged per period whichever CPUs ran);
    measured on the GPU box: north-star step 68-69 ms without affinity, 61-63 ms inside ANY block of 16 CPUs
    (``taskset -c 0-15`` ... ``128-143``, either socket; `lscpu` / `rocm-smi --showtoponuma` on the box).

    ``slot``: which block (one per process of a node: pass the local rank); ``n_cpus``: block size, default 16 (the
    container's CPU quota when that is smaller, at least 8).  An affinity mask that is already narrower than twice the block is
    left alone (the caller -- taskset, numactl, a job scheduler -- has decided); ``MVS_PIN_CPUS=0`` switches this off,
    ``MVS_PIN_CPUS=a-b`` names the CPUs.  Returns the list of CPUs the process may run on afterwards."""
    import os

    if not hasattr(os, "sched_setaffinity"):
        return None
    avail = sorted(os.sched_getaffinity(0))
    env = os.environ.get("MVS_PIN_CPUS", "")
    if env == "0":
        return avail
    if env:
        cpus = set()
        for part in env.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(avail)
    else:
        if n_cpus is None:
            n_cpus = _default_block()
        n_cpus = max(8, min(int(n_cpus), len(avail)))
        if len(avail) < 2 * n_cpus:
            return avail
        first = (int(slot) * n_cpus) % (len(avail) - n_cpus + 1)
        cpus = set(avail[first:first + n_cpus])
    if cpus:
        os.sched_setaffinity(0, cpus)
    return sorted(os.sched_getaffinity(0))


def local_rank_from_env():
    """This process's index among the processes of its node as the common launchers export it, or None."""
    import os

    for key in ("LOCAL_RANK", "SLURM_LOCALID", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID"):
        v = os.environ.get(key, "")
        if v.strip().lstrip("-").isdigit():
            return max(int(v), 0)
    return None


def pin_worker_thread():
    """Initializer of the library's own worker threads (the pair pool of ``compute_pairwise_registrations``): the calling
    THREAD is restricted to the compact CPU block ``pin_process_to_compact_cpus`` would choose for the process -- the
    process itself (the caller's threads) is left alone.  No-op when the process mask is already narrow, when ``MVS_PIN_CPUS=0``,
    and when the environment does not say which process of the node this is (``local_rank_from_env``; ``MVS_PIN_CPUS=a-b`` names
    the CPUs explicitly).  (The batched pair path runs on native worker threads, which inherit the caller's affinity.)"""
    import os
    import threading

    if not hasattr(os, "sched_setaffinity") or os.environ.get("MVS_PIN_CPUS", "") == "0":
        return
    avail = sorted(os.sched_getaffinity(0))
    env = os.environ.get("MVS_PIN_CPUS", "")
    if env:
        cpus = set()
        for part in env.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(avail)
    else:
        # a block is chosen only when the launcher says which process of the node this is (torchrun / SLURM / Open MPI / MVAPICH):
        # processes started without any of these -- dask or joblib workers, several notebooks -- would otherwise all pin their
        # workers to the node's FIRST block.  No rank information: the threads stay where the scheduler puts them.
        slot = local_rank_from_env()
        if slot is None:
            return
        n = max(8, min(_default_block(), len(avail)))
        if len(avail) < 2 * n:
            return
        first = (slot * n) % (len(avail) - n + 1)
        cpus = set(avail[first:first + n])
    try:
        if cpus:
            os.sched_setaffinity(threading.get_native_id(), cpus)
    except OSError:
        pass
