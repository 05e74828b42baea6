"""Can TWO PROCESSES form a multi-process group (bfcg_group_create with a unique id: ncclCommInitRank, one rank per process) on ONE device?
VERDICT r4 item 7 asks for a world-size-2 test of the multi-process protocol before two-GPU hardware shows up, or a precise account of why it
cannot be had.  This runs it: rank 0 makes the RCCL unique id, both processes create their group on device 0.  What RCCL answers is printed
(and committed under profiles/): it refuses two ranks of one communicator on the same device, exactly as it does inside one process
(bfcg_mg.hip switches repeated devices to peer copies there) -- and peer copies need every rank in one process.

    python scripts/probes/rccl_same_device.py            # the parent: spawns both ranks, 90 s limit
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def child(rank, uid_file):
    import bfc_amd
    if rank == 0:
        uid = bfc_amd.GpuGroup.unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            time.sleep(0.05)
            if time.time() - t0 > 30:
                print("rank 1: no unique id after 30 s", flush=True); return
        uid = open(uid_file, "rb").read()
    try:
        g = bfc_amd.GpuGroup(31, 26, [0], max_batch_pos=1 << 20, n_ranks=2, first_rank=rank, uid=uid)
        print("rank %d: group created on device 0: %s" % (rank, g.info()), flush=True)
        g.close()
    except Exception as e:  # noqa: BLE001
        print("rank %d: bfcg_group_create refused: %s" % (rank, e), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2:
        child(int(sys.argv[1]), sys.argv[2])
        sys.exit(0)
    d = tempfile.mkdtemp()
    uid_file = os.path.join(d, "uid")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG=os.environ.get("NCCL_DEBUG", "WARN"))
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), uid_file], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(2)]
    t0 = time.time()
    for r, p in enumerate(ps):
        try:
            out, _ = p.communicate(timeout=max(1, 90 - (time.time() - t0)))
            print("---- rank %d (exit code %d)\n%s" % (r, p.returncode, out[-3000:]))
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
            print("---- rank %d: no answer within 90 s (killed)\n%s" % (r, (out or "")[-3000:]))
