#!/bin/bash
# developer probe: the intermittent silent abort() of the HIP runtime in sessions that share device memory between processes / run out of memory
# (tests/conftest.py: isolated).  Runs the un-isolated tests inline, N sessions, Python-level traceback of any fatal signal kept; odd sessions
# with the runtime's error-level log (AMD_LOG_LEVEL=1).
N=${1:-15}; shift
SEL=${@:-tests}
mkdir -p gpurun_out/abort_hunt
for i in $(seq 1 $N); do
  if [ $((i % 2)) -eq 1 ]; then export AMD_LOG_LEVEL=1; else unset AMD_LOG_LEVEL; fi
  OCEAN_TEST_CHILD=1 timeout 900 python3 -X faulthandler -m pytest $SEL -m gpu -x -q -p no:cacheprovider > gpurun_out/abort_hunt/s$i.out 2> gpurun_out/abort_hunt/s$i.err
  rc=$?
  echo "session $i rc=$rc log=${AMD_LOG_LEVEL:-0} $(tail -1 gpurun_out/abort_hunt/s$i.out | cut -c1-100)"
  if [ $rc -ne 0 ]; then echo "---- stderr tail"; tail -60 gpurun_out/abort_hunt/s$i.err | cut -c1-300; echo "---- stdout tail"; tail -15 gpurun_out/abort_hunt/s$i.out | cut -c1-300; fi
done
