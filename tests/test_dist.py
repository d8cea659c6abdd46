"""Multi-process CPU test of the N>1 path: template-bank shards -> all-gather of the
per-shard match lists -> merge == unsharded result.  gloo backend, world sizes 2 and 3."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_gloo(world, lm, orc):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for k in range(world):
        assert "RANK %d OK" % k in r.stdout


def test_shard_ranges_partition(lm):
    distmod = __import__("importlib").import_module("line-mod-pipeline_amd.dist")
    for n in (0, 1, 7, 3000, 24300):
        for R in (1, 2, 3, 4, 8):
            ranges = [distmod.shard_range(n, r, R) for r in range(R)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(R - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_single_rank_gather_is_identity(lm):
    distmod = __import__("importlib").import_module("line-mod-pipeline_amd.dist")
    g = distmod.ShardGather(lm.merge_matches, cap=16)
    rec = np.zeros((2, 16), lm.MATCH_DTYPE)
    rec["x"][0, :3] = [1, 2, 3]
    out = g.gather_merge(rec, np.array([3, 0], np.int32))
    assert len(out) == 2 and len(out[0]) == 3 and len(out[1]) == 0


def test_pack_and_merge_batch_equal_per_frame_merge(lm):
    """lm_pack_matches / lm_merge_batch (one call per step) against lm_merge_matches per frame."""
    rng = np.random.default_rng(5)
    R, B, cap = 3, 20, 64

    def shard_lists():
        rec = np.zeros((B, cap), lm.MATCH_DTYPE)
        cnt = rng.integers(0, 40, B).astype(np.int32)
        cnt[3] = 0
        for i in range(B):
            m = np.zeros(cnt[i], lm.MATCH_DTYPE)
            m["x"] = rng.integers(0, 6, cnt[i]); m["y"] = rng.integers(0, 6, cnt[i])
            m["similarity"] = rng.integers(80, 84, cnt[i]).astype(np.float32)
            m["template_id"] = rng.integers(0, 5, cnt[i]); m["class_idx"] = 0
            u = lm.merge_matches([m])                           # sorted + unique, as a detector returns it
            cnt[i] = len(u)
            rec[i, :cnt[i]] = u
        return rec, cnt

    shards = [shard_lists() for _ in range(R)]
    packs = [lm.pack_matches(r, c) for r, c in shards]
    for (r, c), p in zip(shards, packs):
        assert len(p) == c.sum() and p[:c[0]].tobytes() == r[0, :c[0]].tobytes()
    stride = max(len(p) for p in packs) + 7
    allr = np.zeros((R, stride), lm.MATCH_DTYPE)
    for k, p in enumerate(packs):
        allr[k, :len(p)] = p
    allc = np.stack([c for _, c in shards])
    merged, mc = lm.merge_batch(allr, allc)
    pos = 0
    for i in range(B):
        exp = lm.merge_matches([shards[k][0][i, :shards[k][1][i]] for k in range(R)])
        assert mc[i] == len(exp) and merged[pos:pos + mc[i]].tobytes() == exp.tobytes()
        pos += mc[i]
    assert pos == len(merged)


@pytest.mark.parametrize("world", [2, 4])
def test_rendezvous_broadcast_multiprocess(world, lm):
    """The TCP rendezvous lm_comm_init uses for the ncclUniqueId (rank 0 -> all ranks), as `world` processes on the
    CPU: every rank must end up with rank 0's 128 bytes; ranks start in arbitrary order (clients retry)."""
    port = _free_port()
    code = (
        "import importlib, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "lm = importlib.import_module('line-mod-pipeline_amd')\n"
        "rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])\n"
        "time.sleep(0.3 * ((rank * 7) %% 3))\n"
        "payload = bytes((i * 37 + 11) %% 256 for i in range(128)) if rank == 0 else bytes(128)\n"
        "for rnd in range(2):\n"                      # two rounds on port, port + 1 like the two lanes' communicators
        "    out = lm.rendezvous_broadcast(rank, world, payload, port=port + rnd, timeout_s=30)\n"
        "    assert out == bytes((i * 37 + 11) %% 256 for i in range(128)), out[:8]\n"
        "print('RANK %%d OK' %% rank)\n" % ROOT)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(world), str(port)], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True, cwd=ROOT) for r in range(world)]
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and "RANK %d OK" % r in out, err[-2000:]


def test_rendezvous_ignores_stray_connections(lm):
    """rank 0 must not abort the rendezvous because something that is not a peer connects (port scanner, a
    stale process of an earlier run): it drops the connection and keeps accepting under its one deadline."""
    import time
    port = _free_port()
    code = (
        "import importlib, sys\n"
        "sys.path.insert(0, %r)\n"
        "lm = importlib.import_module('line-mod-pipeline_amd')\n"
        "rank, port = int(sys.argv[1]), int(sys.argv[2])\n"
        "payload = bytes(range(128)) if rank == 0 else bytes(128)\n"
        "out = lm.rendezvous_broadcast(rank, 2, payload, port=port, timeout_s=30)\n"
        "assert out == bytes(range(128))\n"
        "print('RANK %%d OK' %% rank)\n" % ROOT)
    p0 = subprocess.Popen([sys.executable, "-c", code, "0", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    # strays: one that says nothing and hangs up, one that sends garbage, one that claims to be rank 0 / out of range
    deadline = time.time() + 20
    sent = 0
    while sent < 3 and time.time() < deadline:
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1)
        except OSError:
            time.sleep(0.1)
            continue
        if sent == 1:
            s.sendall(b"GET / HTTP/1.0\r\n\r\n")
        elif sent == 2:
            s.sendall((0x4C4D5256).to_bytes(4, "little") + (7).to_bytes(4, "little"))
        s.close()
        sent += 1
    assert sent == 3
    p1 = subprocess.Popen([sys.executable, "-c", code, "1", str(port)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    for r, p in enumerate((p0, p1)):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0 and "RANK %d OK" % r in out, err[-2000:]


def test_rendezvous_times_out_under_one_deadline(lm):
    """A peer that never arrives: rank 0 gives up after timeout_s (one overall deadline), with a message that says how
    many peers it served."""
    import time
    t0 = time.time()
    with pytest.raises(lm.LinemodError) as e:
        lm.rendezvous_broadcast(0, 3, bytes(128), port=_free_port(), timeout_s=2)
    assert time.time() - t0 < 10 and "0 of 2 peers" in str(e.value)
    t0 = time.time()
    with pytest.raises(lm.LinemodError):
        lm.rendezvous_broadcast(1, 2, bytes(128), port=_free_port(), timeout_s=2)     # nobody listens
    assert time.time() - t0 < 10
