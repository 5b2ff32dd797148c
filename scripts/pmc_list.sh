#!/bin/bash
# the SQ / TCP / TCC counters rocprofv3 offers on the box
cd /tmp && export TMPDIR=/tmp
timeout -k 10 60 rocprofv3 -L 2>&1 | grep -oE "\b(SQ|TCP|TCC|TA|TD|GRBM)_[A-Za-z0-9_]+" | sort -u
