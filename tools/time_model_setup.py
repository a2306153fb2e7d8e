"""Where the set-up time of a variant caller goes (debug aid): checkpoint load, model creation, first forward (workspace
allocation), second forward.   python tools/time_model_setup.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pepper_amd import synthetic  # noqa: E402


def main():
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    tmp = tempfile.mkdtemp()
    sd = synthetic.variant_state_dict(seed=0)
    path = os.path.join(tmp, "m.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), path)
    from pepper_amd.variant.Options import ImageSizeOptions
    from pepper_amd.variant.models.ModelHander import ModelHandler
    from pepper_amd import parallel
    t = [time.perf_counter()]
    state, meta = parallel.load_checkpoint_state(path)
    t.append(time.perf_counter())
    model = ModelHandler.get_new_gru_model(ImageSizeOptions.IMAGE_HEIGHT, meta["gru_layers"], meta["hidden_size"],
                                           ImageSizeOptions.TOTAL_LABELS, ImageSizeOptions.TOTAL_TYPE_LABELS)
    model.load_state_dict(state)
    t.append(time.perf_counter())
    x = torch.from_numpy(np.resize(synthetic.variant_windows(2048, seed=1), (65536, 33, 26))).pin_memory()
    t.append(time.perf_counter())
    model(x, False)
    t.append(time.perf_counter())
    model(x, False)
    t.append(time.perf_counter())
    names = ["torch.load + state_dict", "create (pack + upload)", "(input prep)", "first forward 65536 (workspace alloc)", "second forward"]
    for n, a, b in zip(names, t, t[1:]):
        print("%-40s %7.1f ms" % (n, (b - a) * 1e3))
    t0 = time.perf_counter()
    m2 = ModelHandler.load_simple_model_for_training(path, image_features=ImageSizeOptions.IMAGE_HEIGHT,
                                                     num_classes=ImageSizeOptions.TOTAL_LABELS,
                                                     num_type_classes=ImageSizeOptions.TOTAL_TYPE_LABELS)[0]
    print("%-40s %7.1f ms" % ("ModelHandler.load_simple_model_for_training", (time.perf_counter() - t0) * 1e3))


if __name__ == "__main__":
    main()
