rm -f results.txt checkpoints/* logs/*/*
