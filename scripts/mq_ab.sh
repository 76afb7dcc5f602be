# short-read shapes: automatic geometry (multi-query work-groups) vs the tuning key mq=0, interleaved
A="python scripts/ab.py"
for s in reads50 reads70 reads100 reads130 reads150 reads250; do echo $s; $A $s "" "mq=0" 2>&1 | grep scan; done
