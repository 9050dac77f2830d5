"""Config surface of STOSA-ADT: get_lambdas (stosa/utils.py:376-388) and set_template (:390-396)."""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def get_lambdas(dataset, tp=-1):
    if dataset == "Office":
        return [1e-05, 0, 0.0008], [0.0022, 0.0056, 0.0006]
    if dataset == "Tools":
        return [0, 0.0002, 0.0052], [0.0005, 0.0009, 0.0051]
    if dataset == "Toys":
        return [0.0096, 0, 0.0007], [0.0013, 0, 0.0001]
    if dataset == "Beauty":
        return [0.0021, 0.0068, 0.0005], [0.0009, 0.0066, 0.0094]
    if dataset == "Home":
        return [0.00010069411089658844, 0.009999999997500002, 3.731464248236788e-05], [0.00015787356250004648, 0.000851136830980773, 7.281280851300642e-07]
    raise NotImplementedError("Not supported yes")


def set_template(args, template_folder=None):
    folder = template_folder or os.path.join(_HERE, "templates")
    with open(os.path.join(folder, "%s.json" % args.dataset)) as f:
        for k, v in json.load(f).items():
            setattr(args, k, v)
    return args
