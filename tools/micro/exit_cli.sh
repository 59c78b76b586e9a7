# the cold command line, runs two seconds apart (tools/cli_cold.py --walls)
Q=/tmp/wl_files/human_s0.02_q1
P=/tmp/wl_files/human_s1_q1
ls $Q.bam >/dev/null 2>&1 || python tools/ingest_ab.py --scale 0.02 --seq-mode 1 --configs "d:" --runs 1 --rounds 1 > /dev/null 2>&1
ls $P.bam >/dev/null 2>&1 || python tools/ingest_ab.py --scale 1 --seq-mode 1 --configs "d:" --runs 1 --rounds 1 > /dev/null 2>&1
for k in 1 2 3 4 5 6 7 8; do
sleep 2; echo "200M $(python tools/cli_cold.py --walls $P.bam $P.bed $P.gff 1)"
sleep 1; echo "  4M $(python tools/cli_cold.py --walls $Q.bam $Q.bed $Q.gff 1)"
done
