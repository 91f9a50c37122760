python GA3C.py PLAY_MODE=True "$@"
