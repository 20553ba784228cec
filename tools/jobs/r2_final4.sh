#!/bin/bash
set -u
bash tools/jobs/r2_job33.sh
