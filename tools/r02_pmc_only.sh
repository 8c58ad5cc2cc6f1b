set -e
sed -n '/^SQ1=/,$p' tools/r02_final.sh > /tmp/pmc_part.sh
bash /tmp/pmc_part.sh
