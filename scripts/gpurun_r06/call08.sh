#!/bin/bash
# K3 evidence for the two-lane score pass: kernel stats and SQ counters (separate passes)
bash scripts/collect_profiles_r06.sh k3 2>&1 | tail -30
