"""Config surface of BERT4Rec-ADT: get_lambda (bert4rec/utils.py:263-274) and set_template (:240-250)."""
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def get_lambda(dataset, tp=-1):
    """(lambda_1 reconstruction, lambda_2 independence) per layer: the table of bert4rec/utils.py:263-274."""
    if dataset == "ml-1m":
        return [0.001033064113633401, 5.277219708128945e-06], [0.000899362502660037, 0.000706016178174784]
    if dataset in ("beauty", "Beauty"):
        return [1.4616741512829565e-05, 0.001839446918736823], [0.00037889972403308536, 0.0009180599125696732]
    if dataset == "steam":
        return [0.0003957887657578212, 6.360759018525728e-05], [0.0010088509057684678, 0.0008035241708960854]
    if dataset == "ml-20m":
        return [0.005435293808249262, 0.0019764407654292064], [0.0007068258408279514, 0.0013811031763964325]
    return None


def set_template(args, template_folder=None):
    """The template JSON silently overrides the command line (as in the reference)."""
    folder = template_folder or os.path.join(_HERE, "templates")
    path = os.path.join(folder, "%s.json" % args.dataset)
    if os.path.exists(path):
        with open(path) as f:
            for k, v in json.load(f).items():
                setattr(args, k, v)
    return args
