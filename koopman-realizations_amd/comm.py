"""Multi-GPU plumbing, torch-free: one process per GPU, the collectives are the kp_comm_* entry points of
libkoopman_hip.so (RCCL over xGMI, loaded by the library on first use).

The reference has no parallel construct; its sweeps are serial MATLAB loops over independent units
(Ksysid.train_models over a lasso vector, Ksysid.m:1372-1387; evaluate_rand_models.m:45-144 over random systems).
Here the units are dealt round-robin to ranks (sweep.shard_units), there is no collective on the data path, and the
objects below carry the final gather.  Anything with `.rank`, `.world`, `all_gather_bytes`, `all_reduce_sum` can stand
in for `RcclComm` (the CPU tests pass a gloo-backed object with the same four members).
"""
from __future__ import annotations

import ctypes as C
import os
import pickle
import subprocess
import sys
import tempfile
import time

import numpy as np

from . import _ffi as F


class LocalComm:
    """world = 1: the gather of one."""
    kind = "local"
    rank, world = 0, 1

    def all_gather_bytes(self, payload: bytes):
        return [bytes(payload)]

    def all_reduce_sum(self, a):
        return np.array(a, dtype=np.float64)

    def barrier(self):
        pass


def single_node_defaults():
    """RCCL reads its environment when the library initialises.  All ranks of a launch sit on ONE node here (the xGMI
    mesh), so - unless the caller chose otherwise - the bootstrap sockets use the loopback interface and the InfiniBand
    probe is skipped: an interface that cannot reach itself on a sandboxed box makes ncclCommInitRank wait for ever.
    A launch that names another master address keeps RCCL's own interface search."""
    if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        os.environ.setdefault("NCCL_IB_DISABLE", "1")


class RcclInitTimeout(TimeoutError):
    pass


class RcclComm:
    """kp_comm_create on a Context: RCCL communicator of `world` processes, this one being `rank`.  The call runs under a
    watchdog (KP_COMM_INIT_TIMEOUT seconds, default 240): a bootstrap that never completes raises RcclInitTimeout instead
    of blocking the launch (the blocked thread is abandoned; `init_from_env` then agrees on the file backend)."""
    kind = "rccl"

    def __init__(self, ctx, rank: int, world: int, unique_id: bytes, timeout: float | None = None):
        if len(unique_id) != 128:
            raise ValueError("RCCL unique id is 128 bytes")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        single_node_defaults()
        buf = C.create_string_buffer(unique_id, 128)
        if timeout is None:
            timeout = float(os.environ.get("KP_COMM_INIT_TIMEOUT", "240"))
        box = {}

        def run():
            try:
                box["rc"] = F.lib().kp_comm_create(ctx.handle, C.cast(buf, C.c_void_p), self.rank, self.world)
            except BaseException as e:                    # noqa: BLE001 - handed to the caller's thread
                box["exc"] = e

        import threading
        th = threading.Thread(target=run, name="kp_comm_create", daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            # the blocked thread stays behind; tell the library that whatever it returns later must not become this
            # context's communicator (the launch goes on with another backend)
            try:
                F.lib().kp_comm_abandon(ctx.handle)
            except Exception:                             # noqa: BLE001 - stand-in libraries of the CPU tests
                pass
            raise RcclInitTimeout(f"rank {rank}: ncclCommInitRank did not return within {timeout:.0f} s")
        if "exc" in box:
            raise box["exc"]
        F.check(box["rc"], ctx.handle)

    def all_gather_bytes(self, payload: bytes):
        """Every rank contributes the same number of bytes; returns the world payloads in rank order."""
        n = len(payload)
        send = C.create_string_buffer(bytes(payload), n)
        recv = C.create_string_buffer(n * self.world)
        F.check(F.lib().kp_comm_allgather(self.ctx.handle, C.cast(send, C.c_void_p), n, C.cast(recv, C.c_void_p)), self.ctx.handle)
        raw = recv.raw
        return [raw[r * n:(r + 1) * n] for r in range(self.world)]

    def all_gather_array_direct(self, a):
        """Equally shaped f64 arrays without intermediate byte strings: the library reads `a` and writes the (world,) +
        a.shape result where numpy holds them."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = np.empty((self.world,) + a.shape)
        F.check(F.lib().kp_comm_allgather(self.ctx.handle, a.ctypes.data_as(C.c_void_p), a.nbytes, out.ctypes.data_as(C.c_void_p)),
                self.ctx.handle)
        return out

    def all_reduce_sum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64).copy()
        F.check(F.lib().kp_comm_allreduce_sum(self.ctx.handle, F.dptr(a), a.size), self.ctx.handle)
        return a

    def barrier(self):
        self.all_reduce_sum(np.zeros(1))

    def all_gather_fits(self, first: int, count: int, W: int):
        """The K stacks (fits first .. first + count - 1, kp_fit_get_K numbering) of every rank: ONE RCCL all-gather from the
        device result buffer and one DMA into a page-locked block of the context; (world, count, W, W), blocks column-major."""
        out = self.ctx.host_array("Kgather", (self.world, int(count), int(W), int(W)))
        F.check(F.lib().kp_comm_allgather_fits(self.ctx.handle, int(first), int(count), int(W), F.dptr(out)), self.ctx.handle)
        return out

    def gather_fits(self, root: int, first: int, count: int, W: int):
        """all_gather_fits to ONE rank (kp_comm_gather_fits: ncclSend / ncclRecv, only the root copies the stack to its host).
        Returns the (world, count, W, W) block on the root, None elsewhere."""
        if self.rank == int(root):
            out = self.ctx.host_array("Kgather", (self.world, int(count), int(W), int(W)))
            F.check(F.lib().kp_comm_gather_fits(self.ctx.handle, int(root), int(first), int(count), int(W), F.dptr(out)), self.ctx.handle)
            return out
        F.check(F.lib().kp_comm_gather_fits(self.ctx.handle, int(root), int(first), int(count), int(W), None), self.ctx.handle)
        return None

    def all_gather_fit(self, index: int, W: int):
        """K of fit `index` (kp_fit_get_K numbering) of every rank, gathered device to device: (world, W, W)."""
        K = np.zeros((self.world, W, W))
        F.check(F.lib().kp_comm_allgather_fit(self.ctx.handle, int(index), int(W), F.dptr(K)), self.ctx.handle)
        return np.transpose(K, (0, 2, 1))                 # each block was column-major

    def close(self):
        F.lib().kp_comm_destroy(self.ctx.handle)


class FileComm:
    """Debugging stand-in for RcclComm on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device): the
    same four members, collectives through files in a directory all ranks share.  Selected with KP_COMM_BACKEND=file;
    never the measured path.  `init_from_env` also falls back to it - on every rank, by agreement - when the RCCL
    communicator cannot be created (`fallback_reason` says why), so that a launch still completes and says so."""
    kind = "file"
    fallback_reason = ""

    def __init__(self, ctx, rank: int, world: int, directory: str):
        self.ctx, self.rank, self.world, self.dir, self.seq = ctx, int(rank), int(world), directory, 0
        os.makedirs(directory, exist_ok=True)

    def all_gather_bytes(self, payload: bytes, timeout: float = 600.0):
        self.seq += 1
        mine = os.path.join(self.dir, f"{self.seq}_{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            f.write(bytes(payload))
        os.replace(mine + ".tmp", mine)
        out = []
        t0 = time.time()
        for r in range(self.world):
            path = os.path.join(self.dir, f"{self.seq}_{r}")
            while not os.path.exists(path):
                if time.time() - t0 > timeout:
                    raise TimeoutError(f"FileComm: rank {r} never wrote step {self.seq}")
                time.sleep(0.001)
            with open(path, "rb") as f:
                out.append(f.read())
        return out

    def all_reduce_sum(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return all_gather_array(self, a).sum(axis=0)

    def barrier(self):
        self.all_gather_bytes(b"x")

    def all_gather_fit(self, index: int, W: int):
        return all_gather_array(self, np.ascontiguousarray(self.ctx.fit_result(index, W)))

    def close(self):
        pass


# ---- generic collectives on top of all_gather_bytes ------------------------------------------------------------

def all_gather_array(comm, a):
    """Equally shaped f64 arrays: returns (world,) + a.shape."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    direct = getattr(comm, "all_gather_array_direct", None)
    if direct is not None:
        return direct(a)
    parts = comm.all_gather_bytes(a.tobytes())
    return np.stack([np.frombuffer(p, dtype=np.float64).reshape(a.shape) for p in parts])


def all_gather_fits(comm, ctx, first: int, count: int, W: int, have: int | None = None, root: int | None = None):
    """K stacks of a sharded sweep, (world, count, W, W) with column-major blocks: fits first .. first + count - 1 of every
    rank's device result buffer (`have` = how many of them this rank really computed; the rest is padding).  RCCL: one
    device-to-device all-gather + one DMA (kp_comm_allgather_fits); one rank: the DMA alone; any other `comm` (file / gloo
    stand-ins): the rank's stack through a page-locked block, then the stand-in's array gather.  `root`: only that rank
    receives (and returns) the stack, the others return None (RCCL: kp_comm_gather_fits; stand-ins gather everywhere and drop it)."""
    fn = getattr(comm, "gather_fits" if root is not None else "all_gather_fits", None)
    if fn is not None:
        return fn(root, first, count, W) if root is not None else fn(first, count, W)
    if root is not None and comm is not None and comm.world > 1:
        out = all_gather_fits(comm, ctx, first, count, W, have)
        return out if comm.rank == int(root) else None
    have = count if have is None else min(int(have), int(count))
    if comm is None or comm.world == 1:
        out = ctx.host_array("Kgather", (1, int(count), int(W), int(W)))
        F.check(F.lib().kp_comm_allgather_fits(ctx.handle, int(first), int(count), int(W), F.dptr(out)), ctx.handle)
        return out
    mine = np.zeros((int(count), W, W))
    if have:
        mine[:have] = ctx.fit_results(first, have, W)
    return all_gather_array(comm, mine)


def all_gather_object(comm, obj):
    """Picklable objects of any size: sizes first, then payloads padded to the largest."""
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sizes = [int(np.frombuffer(p, dtype=np.int64)[0]) for p in comm.all_gather_bytes(np.array([len(blob)], dtype=np.int64).tobytes())]
    mx = max(sizes)
    parts = comm.all_gather_bytes(blob + b"\0" * (mx - len(blob)))
    return [pickle.loads(p[:s]) for p, s in zip(parts, sizes)]


def max_over_ranks(comm, x: float) -> float:
    return float(all_gather_array(comm, np.array([x])).max())


# ---- rendezvous of the 128-byte RCCL id (single node: a file) -----------------------------------------------------

def unique_id() -> bytes:
    single_node_defaults()
    buf = C.create_string_buffer(128)
    F.check(F.lib().kp_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


def rendezvous_file_from_env() -> str:
    """Path every rank of one launch agrees on.  KP_COMM_FILE when the launcher set it; otherwise derived from what a
    torch.distributed.run launch exports to all its ranks (MASTER_PORT, run id) plus the launcher's pid."""
    p = os.environ.get("KP_COMM_FILE")
    if p:
        return p
    tag = "_".join([os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"), str(os.getppid())])
    return os.path.join(tempfile.gettempdir(), f"kp_comm_{tag}.id")


LAUNCH_NONCE_SEP = b"|nonce|"


def launch_nonce(path: str) -> str:
    """The 16 hex characters rank 0 wrote behind the id: the rendezvous directory of the backend vote is named after them,
    so a directory left behind by an earlier launch with the same KP_COMM_FILE is never read."""
    try:
        with open(path, "rb") as f:
            raw = f.read()
        if raw[-len(LAUNCH_NONCE_SEP) - 16:-16] == LAUNCH_NONCE_SEP:
            return raw[-16:].decode()
    except OSError:
        pass
    return "0"


def exchange_unique_id(rank: int, path: str, timeout: float = 300.0) -> bytes:
    """Rank 0 creates the id and publishes it atomically (write + rename); the others wait for the file."""
    if rank == 0:
        try:
            uid = unique_id()
        except Exception:
            uid = b"FAIL"                               # RCCL not loadable: tell the other ranks instead of letting them time out
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid + LAUNCH_NONCE_SEP + os.urandom(8).hex().encode())     # the nonce names this launch's vote directory
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            if os.path.getmtime(path) >= t0 - 60.0:     # a file left behind by a crashed earlier launch is not ours
                with open(path, "rb") as f:
                    uid = f.read()
                uid = uid[:-len(LAUNCH_NONCE_SEP) - 16] if uid[-len(LAUNCH_NONCE_SEP) - 16:-16] == LAUNCH_NONCE_SEP else uid
                if len(uid) == 128 or uid == b"FAIL":
                    return uid
        except FileNotFoundError:
            pass
        time.sleep(0.02)
    raise TimeoutError(f"rank {rank}: no RCCL id appeared at {path}")


def init_from_env(Context):
    """One process per GPU: RANK / LOCAL_RANK / WORLD_SIZE from the environment (torch.distributed.run exports them; so
    does `spawn_ranks`).  Returns (ctx, comm); world 1 gives a LocalComm."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if "KP_FORCE_DEVICE" in os.environ:          # debugging on a one-GPU box: all ranks on one device (if RCCL accepts it)
        local = int(os.environ["KP_FORCE_DEVICE"])
    ndev = C.c_int(0)
    F.lib().kp_device_count(C.byref(ndev))
    if 0 < ndev.value <= local:                  # the launcher restricted this rank's visible devices (e.g. one per rank)
        local %= ndev.value
    ctx = Context(local)
    if world == 1:
        return ctx, LocalComm()
    path = rendezvous_file_from_env()
    if os.environ.get("KP_COMM_BACKEND") == "file":      # debugging only: see FileComm
        return ctx, FileComm(ctx, rank, world, path + ".d")
    uid = exchange_unique_id(rank, path)
    comm, why = None, ""
    if len(uid) == 128:
        try:
            comm = RcclComm(ctx, rank, world, uid)
        except Exception as e:                           # e.g. a librccl that does not match the HIP runtime
            why = f"rank {rank}: {e}"
    else:
        why = "kp_comm_unique_id failed on rank 0 (librccl not loadable)"
    # all ranks agree on the backend through the rendezvous directory: one rank without a communicator would leave the
    # others blocked in their first collective
    vote_dir = f"{path}.{launch_nonce(path)}.d"
    fc = FileComm(ctx, rank, world, vote_dir)
    votes = fc.all_gather_bytes((why or "ok").encode()[:200].ljust(200))
    bad = [v.decode().strip() for v in votes if v.decode().strip() != "ok"]
    if bad:
        if comm is not None:
            comm.close()
        fc.fallback_reason = bad[0]
        return ctx, fc
    comm.barrier()
    if rank == 0:
        import shutil
        shutil.rmtree(vote_dir, ignore_errors=True)
        try:
            os.remove(path)
        except OSError:
            pass
    return ctx, comm


def spawn_ranks(argv, world: int, env=None, timeout=None):
    """Starts `world` fresh processes of `argv` (one per GPU) BEFORE anything in this process has touched a GPU and waits
    for them.  Returns rank 0's stdout; raises if a rank failed."""
    fd, path = tempfile.mkstemp(prefix="kp_comm_", suffix=".id")
    os.close(fd); os.remove(path)
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "KP_COMM_FILE": path,
                  "HSA_ENABLE_IPC_MODE_LEGACY": e.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
        procs.append(subprocess.Popen(argv, env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL,
                                      stderr=None if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate(timeout=timeout)
    codes = [procs[0].returncode] + [p.wait(timeout=timeout) for p in procs[1:]]
    if any(codes):
        raise RuntimeError(f"spawn_ranks: exit codes {codes}\n{out}")
    return out
