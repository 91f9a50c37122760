import json, sys
d = json.load(open(sys.argv[1])); e = d['e2e']
print(sys.argv[1].split('/')[-1], 'e2e', round(e['predictions_per_sec']), 'x2', round(e['agents_x2']['predictions_per_sec']), 'x2dev', round(e['agents_x2_frame_queue_on_device']['predictions_per_sec']), 'batch', round(e['mean_predict_batch'], 1))
