# the cold command line with and without the mapping's page-table entries dropped beside the decode (tools/cli_cold.py --walls), runs two seconds apart
P=/tmp/wl_files/human_s1_q1
ls $P.bam >/dev/null 2>&1 || python tools/ingest_ab.py --scale 1 --seq-mode 1 --configs "d:" --runs 1 --rounds 1 > /dev/null 2>&1
for k in 1 2 3 4 5 6; do
sleep 2; echo "dropped $(python tools/cli_cold.py --walls $P.bam $P.bed $P.gff 1)"
sleep 2; echo "kept    $(SPL_KEEP_MAPPING_TABLES=1 python tools/cli_cold.py --walls $P.bam $P.bed $P.gff 1)"
done
