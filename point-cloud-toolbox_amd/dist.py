"""Point-index-range sharding of the curvature path over up to 8 GPUs -- no PyTorch.

The reference is single-process (nothing to mirror); queries are independent, so rank r owns the
contiguous index range ``[r*N/G, (r+1)*N/G)`` of the cloud.  The only exchange is one all-gather of the float32
coordinate shards (12 B/point): every GPU receives the whole candidate set and keeps the points near its own range
(the halo; for an unsorted cloud that is everything).  Outputs need no collective: each rank keeps its K/H rows,
neighbour indices are global, and results do not depend on G.

One process per GPU.  Any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT will do
(``python -m torch.distributed.run`` is used as exactly that and nothing else).  The collective is RCCL over xGMI,
called from ``libpct_hip.so`` (``pct_comm_*``, include/pct_hip.h) on the handle's exchange stream; rank 0 creates the
RCCL unique id and :func:`rendezvous_unique_id` hands its 128 bytes to the other ranks over a TCP socket on
``MASTER_ADDR``, first free port of ``MASTER_PORT+1 .. +8`` (``PCT_RDZV_PORT`` overrides the first candidate).

CPU tests cover ownership, concatenation and the double-buffered step pattern by injecting an exchange object with
the same four methods (``tests/test_dist_gloo.py``: torch.distributed's gloo backend, world size 2) and a checker in
place of the device compute; the rendezvous itself runs there with 2, 4 and 8 processes.
"""
from __future__ import annotations

import os
import socket
import struct
import time

import numpy as np

__all__ = ["shard_range", "shard_sizes", "rendezvous_unique_id", "RcclExchange", "ShardedCurvature", "env_rank"]


def shard_range(n_total, rank, world):
    """Global index range owned by ``rank``."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def shard_sizes(n_total, world):
    return [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]


def env_rank():
    """(rank, world, local_rank) as the launcher exported them; a plain ``python`` run is rank 0 of 1."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


# ------------------------------------------------------------------ rendezvous
def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        buf += chunk
    return bytes(buf)


RDZV_PORTS = 8          # rank 0 listens on the first free port of MASTER_PORT+1 .. +8; peers probe them in turn


def _job_token(addr, port, world):
    """8 bytes that tell this job's rendezvous from another job's on a neighbouring port: both sides derive them from
    what the launcher gave every rank of ONE job (MASTER_ADDR : MASTER_PORT, WORLD_SIZE)."""
    import hashlib
    base = os.environ.get("MASTER_PORT", str(port))
    return hashlib.sha256(f"{addr}:{base}:{world}".encode()).digest()[:8]


def _is_local(addr):
    try:
        ip = socket.gethostbyname(addr)
    except OSError:
        return False
    if ip.startswith("127."):
        return True
    try:
        with socket.socket(socket.AF_INET, socket.SOCK_DGRAM) as s:
            s.bind((ip, 0))
        return True
    except OSError:
        return False


def rendezvous_unique_id(rank, world, make_id, addr=None, port=None, timeout=180.0):
    """Rank 0 calls ``make_id()`` (128 bytes) and serves them to the other ``world - 1`` ranks; every rank returns the
    same bytes.  Each peer announces itself (magic, rank, job token) and the answer carries magic and token back, so
    neither a stray connection, nor a foreign service, nor ANOTHER JOB's rendezvous on one of the candidate ports (two
    jobs on one host whose MASTER_PORTs are within 8 of each other) can be mistaken for this one.  The listener binds
    to MASTER_ADDR when that is an address of this host."""
    if world == 1:
        return make_id()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    if port is None:
        port = int(os.environ.get("PCT_RDZV_PORT", "0")) or int(os.environ.get("MASTER_PORT", "29500")) + 1
    ports = [port + i for i in range(RDZV_PORTS)]
    token = _job_token(addr, port, world)
    deadline = time.monotonic() + timeout
    if rank == 0:
        payload = make_id()
        if len(payload) != 128:
            raise ValueError("the unique id must be 128 bytes")
        srv = None
        for p in ports:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr if _is_local(addr) else "", p))
                s.listen(world)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise OSError(f"rendezvous: none of the ports {ports[0]}..{ports[-1]} is free (set PCT_RDZV_PORT)")
        served = set()
        try:
            while len(served) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise TimeoutError(f"rendezvous: {world - 1 - len(served)} of {world - 1} ranks did not call in") from None
                with conn:
                    conn.settimeout(10.0)
                    try:
                        magic, peer, tok = struct.unpack("!4sI8s", _recv_exact(conn, 16))
                        if magic != b"PCT2" or tok != token or not (0 < peer < world) or peer in served:
                            continue                     # a stray connection, or a rank of another job
                        conn.sendall(b"PCT2" + token + payload)
                    except (ConnectionError, socket.timeout, struct.error, OSError):
                        continue
                    served.add(peer)
        finally:
            srv.close()
        return payload
    last = None
    while time.monotonic() < deadline:
        for p in ports:
            try:
                with socket.create_connection((addr, p), timeout=5.0) as s:
                    s.settimeout(10.0)
                    s.sendall(struct.pack("!4sI8s", b"PCT2", rank, token))
                    answer = _recv_exact(s, 140)
                    if answer[:4] == b"PCT2" and answer[4:12] == token:      # (another job's listener answers nothing we accept)
                        return answer[12:]
            except (ConnectionError, socket.timeout, OSError) as e:     # rank 0 is not listening (there) yet
                last = e
        time.sleep(0.05)
    raise TimeoutError(f"rendezvous: rank {rank} could not reach {addr}:{ports[0]}..{ports[-1]}: {last}")


def scatter_records_host(records, lo, hi):
    """(index bits, K, H) records of every rank -> K, H of rows [lo, hi); the records must cover the range exactly once
    (what ``pct_scatter_records`` checks on the device)."""
    records = np.asarray(records, np.float32).reshape(-1, 3)
    idx = np.ascontiguousarray(records[:, 0]).view(np.int32).astype(np.int64)
    sel = (idx >= lo) & (idx < hi)
    if int(sel.sum()) != hi - lo or len(np.unique(idx[sel])) != hi - lo:
        raise RuntimeError(f"{int(sel.sum())} records for the {hi - lo} rows [{lo},{hi}): the slabs do not partition the cloud")
    K, H = np.empty(hi - lo, np.float32), np.empty(hi - lo, np.float32)
    K[idx[sel] - lo] = records[sel, 1]
    H[idx[sel] - lo] = records[sel, 2]
    return K, H


# ------------------------------------------------------------------ exchange
class RcclExchange:
    """All-gather of device-resident float32 shards through the handle's RCCL communicator (``pct_comm_*``).
    (Multi-process RCCL on this platform needs ``HSA_ENABLE_IPC_MODE_LEGACY=0`` in the environment BEFORE the HIP
    runtime starts -- the host driver supports dmabuf IPC only; ``bench.py`` sets it if the launcher did not.)"""

    def __init__(self, handle, rank, world, addr=None, port=None):
        from . import _capi
        self.handle, self.rank, self.world = handle, int(rank), int(world)
        uid = rendezvous_unique_id(self.rank, self.world, _capi.comm_unique_id, addr, port)
        handle.comm_init(self.rank, self.world, uid)

    def begin(self, send_ptr, recv_ptr, counts):
        """Start gathering ``counts[r]`` floats from every rank r into ``recv_ptr`` (exchange stream)."""
        self.handle.comm_allgather(send_ptr, recv_ptr, counts)
        return recv_ptr

    def end(self, ticket):
        """The handle's compute stream waits for the gather on the device; the host does not block."""
        self.handle.comm_wait()
        return ticket

    def allreduce(self, values, op="sum"):
        return self.handle.comm_allreduce(values, op)

    def barrier(self):
        self.handle.comm_barrier()

    def close(self):
        self.handle.comm_destroy()


class ShardedCurvature:
    """Per-rank driver: all-gather -> neighbour sweep + fit on the owned range.

    GPU ranks: ``handle`` (a ``_capi.Handle``) and ``exchange`` (an :class:`RcclExchange`); the coordinates live in
    device buffers the handle allocates (``setup_device``), two gather buffers alternate so that the exchange of cloud
    i+1 overlaps the kernels of cloud i.  CPU tests inject ``exchange`` (host arrays) and ``compute`` -- a checker with
    the signature ``compute(full_points, lo, hi, k, eps) -> (K, H)``.

    ``ownership``: which rows a rank ANSWERS (it always HOLDS and returns the index range [lo, hi) of the cloud).
    ``"range"``: its own rows -- right when the row order is a spatial order (scan order, tile after tile): the points
    near an index range are a fraction of the cloud.  ``"slab"``: the points of the rank-th of ``world`` slabs of equal
    population along the cloud's longest axis (``pct_set_query_slab``; every rank cuts the gathered cloud alike) --
    for clouds in no spatial order, where the neighbourhood of an index range is the whole cloud and every rank would
    bin every point.  The rows then travel back to their holders in a second exchange: (index, K, H) records, 12 B per
    point, all-gathered (slab populations differ by a few rows: the padded form) and scattered into [lo, hi) on the
    device.  Injected slab checker: ``compute(full_points, rank, world, k, eps) -> (idx, K, H, counts)``.
    """

    def __init__(self, n_total, k, rank, world, eps=None, handle=None, exchange=None, compute=None, ownership="range"):
        self.n_total, self.k, self.rank, self.world, self.eps = int(n_total), int(k), int(rank), int(world), eps
        self.handle, self.exchange, self.compute = handle, exchange, compute
        if ownership not in ("range", "slab"):
            raise ValueError(f"ownership must be 'range' or 'slab', not {ownership!r}")
        self.slab = ownership == "slab" and self.n_total >= 4096      # (pct_set_query_slab's floor; every rank decides alike)
        self._rec_send = self._rec_recv = self._kh = None
        self._rec_cap = 0
        self.lo, self.hi = shard_range(self.n_total, rank, world)
        self.counts = np.asarray(shard_sizes(self.n_total, world), dtype=np.int64) * 3       # floats per rank
        self._send = self._bufs = None
        # a world of one rank issues no collective -- unless PCT_COMM_FORCE asks for one (allgather | padded | bcast):
        # the rehearsal of every form of the exchange on a single GPU
        self.collective = self.world > 1 or (exchange is not None and compute is None and bool(os.environ.get("PCT_COMM_FORCE")))
        if handle is None and compute is None:
            raise RuntimeError("ShardedCurvature needs a device handle (HIP path); there is no CPU fallback")
        if world > 1 and exchange is None:
            raise RuntimeError("more than one rank needs an exchange (RcclExchange on GPUs)")

    # ---- host arrays: the injected path of the CPU tests -------------------------------------------------------
    def step(self, local_pts):
        """One pass on host arrays: exchange + compute.  Returns (K, H) for rows [lo, hi)."""
        local_pts = np.ascontiguousarray(local_pts, dtype=np.float32)
        if local_pts.shape != (self.hi - self.lo, 3):
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo},{self.hi}) of the cloud, got {local_pts.shape}")
        if self.compute is None:
            self.upload_shard(local_pts)
            self._steps = getattr(self, "_steps", -1) + 1           # consecutive steps alternate the gather buffers
            self.run_device(self.end_exchange(self.begin_exchange(self._steps)))
            return self.download()
        full = local_pts if self.world == 1 else self.exchange.allgather_host(local_pts, self.counts)
        if not self.slab:
            return self.compute(full, self.lo, self.hi, self.k, self.eps)
        idx, K, H, counts = self.compute(full, self.rank, self.world, self.k, self.eps)
        if len(idx) != counts[self.rank]:
            raise RuntimeError(f"rank {self.rank}: {len(idx)} slab rows, the cut said {counts[self.rank]}")
        rec = np.empty((len(idx), 3), np.float32)
        rec[:, 0] = np.asarray(idx, np.int32).view(np.float32)
        rec[:, 1], rec[:, 2] = K, H
        if self.world > 1:
            rec = self.exchange.allgather_host(rec.reshape(-1), np.asarray(counts, np.int64) * 3).reshape(-1, 3)
        return scatter_records_host(rec, self.lo, self.hi)

    # ---- device buffers ---------------------------------------------------------------------------------------
    def setup_device(self):
        if self._bufs is None:
            h = self.handle
            self._send = h.device_alloc(max(int(self.counts[self.rank]), 1) * 4)
            self._bufs = [h.device_alloc(max(self.n_total, 1) * 12) for _ in range(2 if self.collective else 1)]
        return self

    def upload_shard(self, local_pts):
        """This rank's rows of the next cloud, into its send buffer (after the previous exchange has read it)."""
        self.setup_device()
        local_pts = np.ascontiguousarray(local_pts, dtype=np.float32)
        if local_pts.shape != (self.hi - self.lo, 3):
            raise ValueError(f"rank {self.rank} must hold rows [{self.lo},{self.hi}) of the cloud, got {local_pts.shape}")
        if self.collective:
            self.handle.comm_synchronize()
        if len(local_pts):
            self.handle.device_upload(self._send if self.collective else self._bufs[0], local_pts)

    def begin_exchange(self, i):
        """Start the all-gather of the resident shard into gather buffer ``i % 2``; returns a ticket."""
        self.setup_device()
        if not self.collective:
            return [None, (self._bufs[0],)]
        self._open = [self.exchange.begin(self._send, self._bufs[i % 2], self.counts), None]
        return self._open

    def end_exchange(self, ticket):
        """The gathered buffer of ``ticket`` (waits for its exchange once; asking again returns the same buffer)."""
        if ticket[1] is None:
            ticket[1] = (self.exchange.end(ticket[0]),)
            if getattr(self, "_open", None) is ticket:
                self._open = None
        return ticket[1][0]

    def run_device(self, full_ptr):
        """Hand the gathered device buffer to the HIP path in place (no copy): it stays untouched until the next
        exchange into it, which is ordered behind this pass on the device (pct_comm_allgather_f32)."""
        h = self.handle
        h.use_points_device(full_ptr, self.n_total)
        if not self.slab:
            h.set_query_range(self.lo, self.hi)
            h.curvature(self.k, self.eps or 0.0)
            return
        h.set_query_slab(self.rank, self.world)
        h.curvature(self.k, self.eps or 0.0)
        counts = np.asarray(h.slab_counts(self.world), dtype=np.int64)
        mine = int(counts[self.rank])
        if self._kh is None:
            own = max(self.hi - self.lo, 1)
            self._kh = (h.device_alloc(own * 4), h.device_alloc(own * 4))
            if self.collective:
                self._rec_recv = h.device_alloc(max(self.n_total, 1) * 12)
        if mine > self._rec_cap:                              # (equal populations: grows once, by the few rows the bins allow)
            if self._rec_send is not None:
                h.synchronize()
                if self.collective:
                    h.comm_synchronize()
                h.device_free(self._rec_send)
            self._rec_cap = min(self.n_total, mine + mine // 16 + 1024)
            self._rec_send = h.device_alloc(self._rec_cap * 12)
        h.slab_records(self._rec_send, self._rec_cap)
        records = self._rec_send
        if self.collective:
            # one exchange at a time per handle: the gather of the NEXT cloud may be in flight on the exchange stream
            # (it overlapped this pass); the records queue up behind it there
            if getattr(self, "_open", None) is not None:
                self.end_exchange(self._open)
            records = self.exchange.end(self.exchange.begin(self._rec_send, self._rec_recv, counts * 3))
        h.scatter_records(records, self.n_total, self.lo, self.hi, *self._kh)

    def download(self):
        if self.slab:
            K, H = np.empty(self.hi - self.lo, np.float32), np.empty(self.hi - self.lo, np.float32)
            if len(K):
                self.handle.device_download(self._kh[0], K)
                self.handle.device_download(self._kh[1], H)
            return K, H
        _, K, H, _ = self.handle.get_fit(self.lo, self.hi, coefs=False, H2=False)
        return K, H

    def close(self):
        if self._bufs is not None:
            h = self.handle
            h.synchronize()
            if self.collective:
                h.comm_synchronize()
            for p in [self._send] + self._bufs + [self._rec_send, self._rec_recv] + list(self._kh or ()):
                if p is not None:
                    h.device_free(p)
            self._send = self._bufs = None
