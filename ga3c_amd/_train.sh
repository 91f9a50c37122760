mkdir -p checkpoints logs
python GA3C.py "$@"
