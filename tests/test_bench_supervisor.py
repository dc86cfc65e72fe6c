"""bench.py --gpus N: every rank supervises a measuring child and the ranks fall back TOGETHER - collectives inside the
hipGraph -> hipGraph segments -> eager launches - when a child dies or stops making progress (the first
real multi-GPU run must not be losable to a hang of the never-measured default path).  Driven here on the CPU with a
stand-in child (DUSTY_BENCH_CHILD_CMD): no GPU, no process group inside the children."""
import json
import os
import socket
import subprocess
import sys
import textwrap
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE = textwrap.dedent('''
    import json, os, sys, time
    rank, a = int(os.environ["RANK"]), int(os.environ["DUSTY_BENCH_ATTEMPT"])
    mode = os.environ.get("FAKE_MODE", "hang")
    with open(os.path.join(os.environ["FAKE_DIR"], f"pid_r{rank}_a{a}"), "w") as fh:
        fh.write(str(os.getpid()))
    with open(os.path.join(os.environ["FAKE_DIR"], f"env_r{rank}_a{a}"), "w") as fh:
        json.dump({k: os.environ.get(k) for k in ("DUSTY_GAN_GRAPH_COMM", "DUSTY_GAN_GRAPH_DDP", "MASTER_PORT",
                                                  "TORCHELASTIC_USE_AGENT_STORE", "DUSTY_BENCH_CHILD")}, fh)
    def progress(s):
        with open(os.environ["DUSTY_BENCH_PROGRESS"], "a") as fh:
            fh.write(s + "\\n")
    progress("started")
    if a == 0 and mode == "hang":
        if rank == 1:
            time.sleep(3600)               # a rank stuck in a captured collective: no progress, no exit
        while True:                        # its peer keeps "waiting for it" (alive, making no progress either)
            time.sleep(0.2)
    if a == 0 and mode == "die" and rank == 0:
        sys.exit(3)                        # a rank that crashes; its peer would wait for it forever
    if a == 0 and mode == "die":
        time.sleep(3600)
    if a == 1 and mode == "die":
        sys.exit(4)                        # the segmented form fails too: the eager form must still be tried
    for i in range(3):
        progress(f"step {i}")
        time.sleep(0.1)
    if rank == 0:
        print("some library chatter on stdout")
        print(json.dumps({"metric": "fake", "value": 1.0, "n_gpus": 2, "distributed": {"world_size": 2}}))
''')


def run_supervisors(tmp_path, mode):
    fake = tmp_path / "fake_child.py"
    fake.write_text(FAKE)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in range(2):
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "DUSTY_BENCH_CHILD_CMD": f"{sys.executable} {fake}",
                    "DUSTY_BENCH_STALL_S": "3", "DUSTY_BENCH_CAP_S": "30", "FAKE_DIR": str(tmp_path), "FAKE_MODE": mode})
        env.pop("DUSTY_BENCH_CHILD", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    return procs, outs


def alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    # (a killed child of a dead supervisor may linger as a zombie of init for a moment: read its state)
    try:
        with open(f"/proc/{pid}/stat") as fh:
            return fh.read().split()[2] != "Z"
    except OSError:
        return False


@pytest.mark.timeout(300)
def test_a_hung_attempt_costs_one_attempt_not_the_record(tmp_path):
    procs, outs = run_supervisors(tmp_path, "hang")
    assert [p.returncode for p in procs] == [0, 0], outs
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]   # ONE line, from rank 0
    rec = json.loads(lines[0])
    att = rec["distributed"]["launch_attempts"]
    assert [a["attempt"] for a in att] == [0, 1] and att[0]["outcome"] != "ok" and att[1]["outcome"] == "ok"
    assert "no progress" in att[0]["outcome"] or "another rank" in att[0]["outcome"]
    assert rec["metric"] == "fake" and rec["distributed"]["world_size"] == 2
    # the hung children are gone (their whole process groups were killed), and the second attempt ran as segments on a
    # rendezvous port of its own with the child hosting its store
    time.sleep(0.5)
    for rank in range(2):
        assert not alive(int((tmp_path / f"pid_r{rank}_a0").read_text()))
        e0 = json.loads((tmp_path / f"env_r{rank}_a0").read_text())
        e1 = json.loads((tmp_path / f"env_r{rank}_a1").read_text())
        assert e0["DUSTY_GAN_GRAPH_COMM"] is None and e1["DUSTY_GAN_GRAPH_COMM"] == "0" and e1["DUSTY_GAN_GRAPH_DDP"] is None
        assert e0["MASTER_PORT"] != e1["MASTER_PORT"] and e1["TORCHELASTIC_USE_AGENT_STORE"] is None
        assert e0["DUSTY_BENCH_CHILD"] == "1"
    assert json.loads((tmp_path / "env_r0_a1").read_text())["MASTER_PORT"] == json.loads((tmp_path / "env_r1_a1").read_text())["MASTER_PORT"]


@pytest.mark.timeout(300)
def test_a_crashed_rank_takes_every_rank_to_the_next_mode(tmp_path):
    procs, outs = run_supervisors(tmp_path, "die")
    assert [p.returncode for p in procs] == [0, 0], outs
    rec = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][0])
    att = rec["distributed"]["launch_attempts"]
    assert [a["outcome"] == "ok" for a in att] == [False, False, True]
    assert "exited 3" in att[0]["outcome"] and "exited 4" in att[1]["outcome"]
    assert att[2]["mode"] == "eager launches"
    e2 = json.loads((tmp_path / "env_r1_a2").read_text())
    assert e2["DUSTY_GAN_GRAPH_COMM"] == "0" and e2["DUSTY_GAN_GRAPH_DDP"] == "0"
    time.sleep(0.5)
    assert not alive(int((tmp_path / "pid_r1_a0").read_text()))
