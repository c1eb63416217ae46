#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for lib in tools/_variant_prof.so tools/_variant_prof_nostore.so; do for dn in "" 0; do echo "== $lib DENSITY=${dn:-scans}"; if [ -z "$dn" ]; then CAELO_LIB=$lib python tools/enc_phase_prof_x.py 2>&1 | tail -10; else DENSITY=$dn CAELO_LIB=$lib python tools/enc_phase_prof_x.py 2>&1 | tail -10; fi; done; done
