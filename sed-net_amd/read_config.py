"""Config -- same attribute bag as /root/reference/read_config.py:6-84, read with a small INI parser (the reference
depends on ConfigObj, absent here). The `.yml` files are ConfigObj-style: top-level `key = value`, one `[train]`
section, `#` comments, optional quotes."""


def _parse(filename):
    top, sections, cur = {}, {}, None
    with open(filename) as fh:
        for raw in fh:
            line = raw.strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("[") and line.endswith("]"):
                cur = sections.setdefault(line[1:-1].strip(), {})
                continue
            if "=" not in line:
                continue
            key, val = line.split("=", 1)
            val = val.strip()
            if val[:1] in "\"'":                       # quoted: keep everything up to the closing quote
                q = val[0]
                end = val.find(q, 1)
                val = val[1:end] if end > 0 else val[1:]
            else:
                val = val.split("#", 1)[0].strip()
            (cur if cur is not None else top)[key.strip()] = val
    return top, sections


def _as_bool(v):
    s = str(v).strip().lower()
    if s in ("true", "yes", "on", "1"):
        return True
    if s in ("false", "no", "off", "0"):
        return False
    raise ValueError(f"not a boolean: {v!r}")


class Config(object):
    def __init__(self, filename):
        self.filename = filename
        top, sections = _parse(filename)
        t = sections["train"]
        self.config = {"comment": top.get("comment", ""), "train": t}
        self.comment = top.get("comment", "")
        self.model_path = t["model_path"]
        self.dataset = t["dataset"]
        self.pretrain_model_path = t["pretrain_model_path"]
        self.pretrain_model_type_path = t["pretrain_model_type_path"]
        self.preload_model = _as_bool(t["preload_model"])
        self.pretrain_opti_path = t["pretrain_opti_path"]
        self.normals = _as_bool(t["normals"])
        self.smooth = float(t["smooth"])
        self.eval_T = int(t["eval_T"])
        self.num_train = int(t["num_train"])
        self.num_val = int(t["num_val"])
        self.num_test = int(t["num_test"])
        self.num_points = int(t["num_points"])
        self.grid_size = int(t["grid_size"])
        self.embed = int(t["embed"])
        self.loss_weight = float(t["loss_weight"])
        self.dataset_path = t["dataset"]
        self.weight_decay = float(t["weight_decay"])
        self.epochs = int(t["num_epochs"])
        self.batch_size = int(t["batch_size"])
        self.gpu = t["gpu"]
        self.mode = int(t["mode"])
        self.lr = float(t["lr"])
        self.patience = int(t["patience"])
        self.optim = t["optim"]
        self.sche = t["sche"]
        self.input_drop = float(t["encoder_drop"])
        self.lr_sch = _as_bool(t["lr_sch"])
        try:
            self.knn = int(t["knn"])
        except (KeyError, ValueError):
            self.knn = 64                                   # read_config.py:80-84
            print("config no knn! use default 64")
