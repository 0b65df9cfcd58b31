#!/bin/bash
for v in 5 6; do
  PT_LDS_SHARE=$v bash tools/dbg/ab.sh "share$v" atrium260k blob7m terrain1m 2>&1 | grep share
done
