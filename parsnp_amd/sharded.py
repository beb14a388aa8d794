"""Sharded --no-partition run over the GPUs of one node: one process per GPU under torch.distributed,
query genomes split into contiguous blocks, the reference on every GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m parsnp_amd.sharded <file.ini>

Every rank ingests the ini's genomes and runs the same host logic (it needs every genome for MUM validation and the
layout); the engine of rank r only keeps block r resident and searches it.  Per engine batch two exchanges run over
torch.distributed (backend nccl = RCCL over xGMI when there is one GPU per rank, gloo otherwise):
  all-reduce(min) of Master.EP, all-gather of the per-genome candidate columns.
Rank 0 writes parsnpAligner.xmfa / .log (or all.mumi).  Results are bit-identical to the single-GPU run."""
import ctypes as C
import json
import os
import sys

import numpy as np

from .core_api import CORE_LIB

AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int64)
AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class ShardedRun:
    def __init__(self, ini_path, dist, device=None, lib_path=None, rccl=False):
        """rccl=True: the engine's own RCCL communicator carries both exchanges on device buffers (pm_session_create_rccl);
        torch.distributed only hands rank 0's communicator id to the other ranks.  rccl=False: the exchanges go through
        the two callbacks below on host buffers (gloo in the CPU tests, or ranks that share a GPU)."""
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dev = device   # torch device for the collectives ("cuda:N" with nccl, "cpu" with gloo)
        L = self.L = C.CDLL(lib_path or CORE_LIB)
        L.pc_step.argtypes = [C.c_void_p]; L.pc_step.restype = C.c_char_p
        L.pc_write.argtypes = [C.c_void_p]; L.pc_mumi.argtypes = [C.c_void_p]; L.pc_close.argtypes = [C.c_void_p]
        if rccl:
            ident = (C.c_uint8 * 128)()
            if self.rank == 0 and L.pc_rccl_id(ident):
                raise RuntimeError("cannot create an RCCL communicator id")
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0)          # 128 bytes through the launcher's channel
            ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
            h = C.c_void_p()
            L.pc_open_rccl.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_void_p)]
            rc = L.pc_open_rccl(ini_path.encode(), self.rank, self.world, ident, C.byref(h))
            if rc:
                raise RuntimeError("parsnp_core could not start (exit code %d)" % rc)
            self.h = h
            return
        L.pc_open_sharded.argtypes = [C.c_char_p, C.c_int, C.c_int, AR, AG, C.c_void_p, C.POINTER(C.c_void_p)]
        L.pc_step.argtypes = [C.c_void_p]; L.pc_step.restype = C.c_char_p
        L.pc_write.argtypes = [C.c_void_p]; L.pc_mumi.argtypes = [C.c_void_p]; L.pc_close.argtypes = [C.c_void_p]
        self._ar, self._ag = AR(self._allreduce_min), AG(self._allgather)   # keep the thunks alive
        h = C.c_void_p()
        rc = L.pc_open_sharded(ini_path.encode(), self.rank, self.world, self._ar, self._ag, None, C.byref(h))
        if rc:
            raise RuntimeError("parsnp_core could not start (exit code %d)" % rc)
        self.h = h

    # ---- the two exchange steps of include/parsnp_mum.h, on host buffers
    def _allreduce_min(self, ctx, buf, count):
        try:
            a = np.ctypeslib.as_array(buf, (count,))
            t = self.torch.from_numpy(a).to(self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
            a[:] = t.cpu().numpy()
            return 0
        except Exception as e:   # never let an exception cross the C boundary
            sys.stderr.write("all-reduce failed: %r\n" % (e,))
            return 1

    def _allgather(self, ctx, send, nbytes, recv):
        try:
            s = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), (nbytes,))
            r = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), (nbytes * self.world,))
            t = self.torch.from_numpy(s.copy()).to(self.dev)
            out = self.torch.empty(nbytes * self.world, dtype=self.torch.uint8, device=self.dev)
            self.dist.all_gather_into_tensor(out, t)
            r[:] = out.cpu().numpy()
            return 0
        except Exception as e:
            sys.stderr.write("all-gather failed: %r\n" % (e,))
            return 1

    def step(self, intervals=True):      # (intervals: CoreRun.step's switch; a sharded run always reports them)
        return json.loads(self.L.pc_step(self.h).decode())

    def mumi(self):
        return self.L.pc_mumi(self.h)

    def write(self):
        return self.L.pc_write(self.h) if self.rank == 0 else 0

    def close(self):
        if self.h:
            self.L.pc_close(self.h)
            self.h = None


def main(argv=None):
    import torch
    import torch.distributed as dist
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        sys.exit("usage: python -m torch.distributed.run ... -m parsnp_amd.sharded <file.ini>")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rccl = False
    if torch.cuda.is_available() and torch.cuda.device_count() >= world and os.environ.get("PARSNP_SHARD_COLLECTIVES", "rccl") == "rccl":
        # one GPU per rank: the engine's own RCCL communicator (device buffers, xGMI); torch.distributed is only the
        # launcher's rendezvous here (gloo), so the process holds ONE RCCL -- the engine's
        os.environ["PARSNP_DEVICE"] = str(local)   # the engine links the system HIP runtime, not torch's
        dist.init_process_group("gloo")
        dev = "cpu"
        rccl = True
    elif torch.cuda.is_available() and torch.cuda.device_count() >= world:
        torch.cuda.set_device(local)
        os.environ["PARSNP_DEVICE"] = str(local)
        dist.init_process_group("nccl")            # PARSNP_SHARD_COLLECTIVES=torch: host-staged exchanges through torch's RCCL
        dev = "cuda:%d" % local
    else:
        if torch.cuda.is_available():
            os.environ["PARSNP_DEVICE"] = str(local % torch.cuda.device_count())
        dist.init_process_group("gloo")
        dev = "cpu"
    lib = os.environ.get("PARSNP_CORE_LIB")   # tests point this at their CPU build
    run = ShardedRun(argv[0], dist, dev, lib, rccl=rccl)
    calcmumi = any(l.strip().lower() == "calcmumi=1" for l in open(argv[0]))
    if calcmumi:
        rc = run.mumi()
    else:
        rep = run.step()
        rc = 0
        if rep["mums_found"]:
            run.write()
        if dist.get_rank() == 0:
            print(json.dumps({k: rep[k] for k in ("anchors", "mums", "lcbs", "core_bp", "path_s", "finder_calls")}))
    dist.barrier()
    run.close()
    dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
