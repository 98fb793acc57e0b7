"""Command line of the MI355X path, shaped like the reference's sacred CLIs
(`python Training.py with cfg.full_44KHz`, `python Predict.py with cfg.full_44KHz input_path=...`;
/root/reference/Training.py:153-166, Predict.py:1-17, Config.py:52-161):

  python -m wave_u_net_amd train   with cfg.baseline_stereo model_config.epoch_it=200 data_root=DATA
  python -m wave_u_net_amd train   with cfg.m1_context synthetic=1 experiment_id=7
  python -m wave_u_net_amd predict with cfg.full model_path=ckpt.npz input_path=mix.wav output_path=out
  python -m wave_u_net_amd test    with cfg.baseline model_path=ckpt.npz data_root=DATA partition=valid
(model_path / load_model: a .npz of this package or a TensorFlow V2 checkpoint prefix such as checkpoints/full_44KHz/full_44KHz-236118)

`with` arguments: `cfg.<named config>` (any of wave_u_net_amd.NAMED_CONFIGS), `model_config.<key>=<value>`
overrides, and the command's own options as `<name>=<value>`.  data_root holds
train|valid|test/<track>/<source>.wav|.npy (+ optional mix.wav) at expected_sr.  Multi-GPU:
launch `train` with `python -m torch.distributed.run --nproc-per-node N -m wave_u_net_amd train with ...`.
"""
import ast
import os
import random
import sys


def _parse(argv):
    if len(argv) < 1 or argv[0] not in ("train", "predict", "test"):
        raise SystemExit(__doc__)
    cmd, rest = argv[0], argv[1:]
    if rest and rest[0] == "with":
        rest = rest[1:]
    name, overrides, opts = "baseline", {}, {}
    for tok in rest:
        if tok.startswith("cfg."):
            name = tok[4:]
        elif "=" in tok:
            key, val = tok.split("=", 1)
            try:
                val = ast.literal_eval(val)
            except (ValueError, SyntaxError):
                pass
            if key.startswith("model_config."):
                overrides[key[len("model_config."):]] = val
            else:
                opts[key] = val
        else:
            raise SystemExit("cannot parse argument %r" % tok)
    return cmd, name, overrides, opts


def main(argv=None):
    cmd, name, overrides, opts = _parse(sys.argv[1:] if argv is None else argv)
    import wave_u_net_amd as wun
    from wave_u_net_amd import training, validation, evaluate
    if name not in wun.NAMED_CONFIGS:
        raise SystemExit("unknown named config %r (have: %s)" % (name, ", ".join(sorted(wun.NAMED_CONFIGS))))
    model_config = wun.get_config(name, **overrides)

    if cmd == "train":
        experiment_id = opts.get("experiment_id", random.randint(0, 1000000))       # Config.py:5 (sacred seed-derived id)
        for d in (model_config["model_base_dir"], model_config["log_dir"]):         # Training.py:158-160
            os.makedirs(d, exist_ok=True)
        if opts.get("synthetic"):
            path = training.train(model_config, experiment_id, load_model=opts.get("load_model"))
            print("Saved model at " + str(path))
        elif opts.get("optimise", True) and "data_root" in opts:
            path, loss = validation.optimise(model_config, experiment_id, data_root=opts["data_root"],
                                             max_epochs=opts.get("max_epochs"))
            print("Supervised training finished! Saved model at " + str(path) + ". Performance: " + str(loss))
        else:
            raise SystemExit("train needs data_root=<dir> or synthetic=1")
    elif cmd == "test":
        loss = validation.test(model_config, opts.get("partition", "test"), str(opts.get("experiment_id", "cli")),
                               opts.get("model_path"), data_root=opts["data_root"])
        print("Finished testing - Mean MSE: " + str(loss))
    else:
        if "input_path" not in opts:
            raise SystemExit("predict needs input_path=<mixture.wav>")
        evaluate.produce_source_estimates(model_config, opts.get("model_path"), opts["input_path"], opts.get("output_path"))


if __name__ == "__main__":
    main()
