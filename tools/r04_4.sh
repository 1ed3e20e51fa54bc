cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04d
python -m pytest tests -m gpu -x -q -k "wide_keys or host_mirror" > gpurun_out/${T}_newtests.log 2>&1; tail -5 gpurun_out/${T}_newtests.log
RTS_LAP=1 python tools/py_begin_probe.py sphere6 2>&1 | tail -5
RTS_LAP=1 python tools/py_begin_probe.py c3 2>&1 | tail -5
