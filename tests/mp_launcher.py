"""Starts rank processes for tests/test_multiprocess_gpu.py.

This helper is started by tests/conftest.py BEFORE the pytest process touches
the GPU and never touches it itself, so the processes it starts are children
of a process without a HIP runtime -- a process that has initialised the GPU
must not fork + exec (the GPU pool's rule), and by the time the multi-process
test runs, pytest has.

Protocol: one JSON object per line on stdin
    {"world": 2, "script": "...", "args": [...], "timeout": 300}
-> starts `world` fresh interpreters ``python script <rank> <world> <port> *args``
and answers with one JSON line {"rcs": [...], "tails": ["last 2000 chars of
each rank's output", ...]}.  EOF ends the helper."""
import json
import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main():
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            req = json.loads(line)
            world, port = int(req["world"]), _free_port()
            env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='1')
            env.update(req.get("env") or {})
            procs = [subprocess.Popen([sys.executable, req["script"], str(r), str(world), str(port)] + list(req["args"]),
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                     for r in range(world)]
            rcs, tails = [], []
            for p in procs:
                try:
                    out, _ = p.communicate(timeout=float(req.get("timeout", 300)))
                except subprocess.TimeoutExpired:
                    p.kill()
                    out, _ = p.communicate()
                    out = (out or '') + '\n[killed: timeout]'
                rcs.append(p.returncode)
                tails.append((out or '')[-2000:])
            ans = {"rcs": rcs, "tails": tails}
        except Exception as exc:
            ans = {"error": repr(exc)}
        sys.stdout.write(json.dumps(ans) + '\n')
        sys.stdout.flush()


if __name__ == '__main__':
    main()
