"""bench.py's N > 1 control flow, rehearsed on ONE GPU (JMHIP_BENCH_REHEARSAL=1: every rank on cuda:0, gloo, exchange through host
memory): torch.distributed.run launches two (and three) ranks, each codes its band of macroblock rows, the bands are gathered every frame.
The reference picture after the last step must be the one a single process produces -- same frames, only sharded -- and only
rank 0 may print the JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-1500:], r.stderr[-1500:])
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])          # 3: uneven bands (23 + 23 + 22 macroblock rows), padded exchange chunks
def test_ranks_rebuild_the_same_reference_as_one(ranks):
    one = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-mbs", "0"])
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(29539 + ranks), "bench.py", "--gpus", str(ranks), "--size", "1080p", "--steps", "3", "--warmup", "1", "--cpu-mbs", "0"],
              {"JMHIP_BENCH_REHEARSAL": "1"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == ranks and two["config"]["slices"] == ranks
    assert one["ref_checksum"] == two["ref_checksum"], "the sharded run did not reproduce the single-process reference picture"
    for d in (one, two):
        for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d


@pytest.mark.gpu
def test_two_ranks_with_the_loop_filter_inside_their_slices():
    """--deblock: every rank filters its own slice (idc 2) before the exchange; one GPU filtering the same two slices must agree."""
    one = run([sys.executable, "bench.py", "--deblock", "--deblock-slices", "2", "--steps", "2", "--warmup", "1", "--cpu-mbs", "0"])
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29547", "bench.py", "--gpus", "2", "--size", "1080p", "--deblock", "--steps", "2", "--warmup", "1", "--cpu-mbs", "0"],
              {"JMHIP_BENCH_REHEARSAL": "1"})
    plain = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--cpu-mbs", "0"])
    assert one["ref_checksum"] == two["ref_checksum"]
    assert one["ref_checksum"] != plain["ref_checksum"], "the filter changed nothing"


@pytest.mark.gpu
def test_rccl_transport_with_a_process_group_of_one():
    """What a one-GPU box can exercise of the RCCL path: JMHIP_BENCH_RCCL1=1 takes bench.py's N > 1 code -- band interpolation, one-chunk
    pack, dist.all_gather_into_tensor over the nccl (= RCCL) backend enqueued on the library's own HIP stream, scatter -- with a process
    group of a single rank. Same reference picture as the plain run."""
    one = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-mbs", "0"])
    r1 = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-mbs", "0"],
             {"JMHIP_BENCH_RCCL1": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29561"})
    assert one["ref_checksum"] == r1["ref_checksum"]
    assert r1["n_gpus"] == 1


@pytest.mark.gpu
def test_two_ranks_default_to_the_4k_configuration():
    """--gpus > 1 without --size runs BASELINE configs[3] (3840x2160): same reference picture as one GPU coding that picture."""
    one = run([sys.executable, "bench.py", "--size", "2160p", "--steps", "2", "--warmup", "1", "--cpu-mbs", "0"])
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29551", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-mbs", "0"], {"JMHIP_BENCH_REHEARSAL": "1"})
    assert "3840x2160" in two["config"]["workload"] and "3840x2160" in one["config"]["workload"]
    assert one["ref_checksum"] == two["ref_checksum"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [2, 3])
def test_ranks_of_the_jm_exact_path_rebuild_the_reference_of_one_gpu_coding_the_same_slices(ranks):
    """bench.py --exact: every rank searches ITS slice with jmhip_p_slice_search (JM's predictors never cross a slice, src/mb_access.c:30-36),
    hands it to the frame stage (jmhip_slice_to_frame_band) and the reconstructed bands are gathered. One GPU coding the same picture cut into
    the same slices (--exact-slices, all slices in one call) must produce the same reference pictures; and cutting the picture differently must
    not (the slices' predictors differ), which shows the slicing is really in the result."""
    one = run([sys.executable, "bench.py", "--exact", "--exact-slices", str(ranks), "--steps", "2", "--warmup", "1"])
    many = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                "--master-port", str(29570 + ranks), "bench.py", "--gpus", str(ranks), "--size", "1080p", "--exact", "--steps", "2", "--warmup", "1"],
               {"JMHIP_BENCH_REHEARSAL": "1"})
    whole = run([sys.executable, "bench.py", "--exact", "--steps", "2", "--warmup", "1"])
    assert many["n_gpus"] == ranks and many["config"]["slices"] == ranks and one["config"]["slices"] == ranks
    assert one["ref_checksum"] == many["ref_checksum"], "the ranks' slices did not reproduce the one-GPU coding of the same slices"
    assert whole["ref_checksum"] != one["ref_checksum"], "one slice and %d slices gave the same picture: the slicing is not in the result" % ranks
    assert "jmhip_p_slice_search" in many["roofline"]["kernel"]
