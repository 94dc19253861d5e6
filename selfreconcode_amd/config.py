"""Configuration access with the pyhocon surface the reference uses (get_int / get_float / get_bool /
get_string / get_config / `in`, dotted keys) plus a small HOCON-subset parser for config.conf-style
files (pyhocon itself is not a dependency).  `default_config()` restates the shipped
config.conf (train / sdf_net / mlp_deformer / render_net / loss_coarse|medium|fine)."""
import re


class Conf(dict):
    def _walk(self, key):
        node = self
        for part in key.split('.'):
            if not isinstance(node, dict) or not dict.__contains__(node, part):
                raise KeyError(key)
            node = node[part]
        return node

    def __contains__(self, key):
        try:
            self._walk(key)
            return True
        except KeyError:
            return False

    def get(self, key, default=None):
        try:
            return self._walk(key)
        except KeyError:
            return default

    def get_float(self, key):
        return float(self._walk(key))

    def get_int(self, key):
        return int(float(self._walk(key)))

    def get_bool(self, key):
        v = self._walk(key)
        return v if isinstance(v, bool) else str(v).lower() == 'true'

    def get_string(self, key):
        return str(self._walk(key))

    def get_config(self, key):
        v = self._walk(key)
        return v if isinstance(v, Conf) else Conf(v)

    def get_list(self, key):
        return list(self._walk(key))


def _scalar(tok):
    tok = tok.strip().strip('"')
    if tok.lower() in ('true', 'false'):
        return tok.lower() == 'true'
    try:
        return int(tok)
    except ValueError:
        try:
            return float(tok)
        except ValueError:
            return tok


def parse_hocon(text):
    """Enough HOCON for the reference's .conf files: nested `name { ... }`, `key = value`, lists in [ ]."""
    toks = re.findall(r'"[^"]*"|[{}\[\]=]|[^\s{}\[\]=]+', re.sub(r'(#|//).*', '', text))
    pos = 0

    def block():
        nonlocal pos
        out = Conf()
        while pos < len(toks) and toks[pos] != '}':
            key = toks[pos].strip('"'); pos += 1
            if toks[pos] == '=':
                pos += 1
            if toks[pos] == '{':
                pos += 1
                out[key] = block()
                pos += 1
            elif toks[pos] == '[':
                pos += 1
                lst = []
                while toks[pos] != ']':
                    lst.append(_scalar(toks[pos].rstrip(','))); pos += 1
                pos += 1
                out[key] = lst
            else:
                out[key] = _scalar(toks[pos]); pos += 1
        return out
    return block()


def load_config(path):
    with open(path) as fh:
        return parse_hocon(fh.read())


def _loss(color, dct, pc_w, lap, defc_w, def_regu_w=0.1, sample_pix=None):
    d = Conf(color_weight=color, normal_weight=0.1, weighted_normal=True, grad_weight=1., offset_weight=0.,
             def_regu=Conf(weight=def_regu_w, c=0.5), dct_weight=dct,
             pc_weight=Conf(weight=pc_w, laplacian_weight=lap, edge_weight=-10., norm_weight=-0.001,
                            def_consistent=Conf(weight=defc_w, c=0.01)))
    if sample_pix is not None:
        d['sample_pix_num'] = sample_pix
    return d


def default_config():
    return Conf(
        train=Conf(nepoch=200, sample_pix_num=2048, initial_iters=-1200, skinner_pose_type=1, shuffle=True, num_workers=4,
                   opt_pose=True, opt_trans=True,
                   opt_camera=Conf(focal_length=True, princeple_points=True, quat=False, T=True),
                   learning_rate=0.0001, scheduler=Conf(type="MultiStepLR", milestones=[10, 30, 80, 130], factor=0.333),
                   coarse=Conf(start_epoch=0, point_render=Conf(radius=0.006, remesh_intersect=30, batch_size=3)),
                   medium=Conf(start_epoch=6, point_render=Conf(radius=0.00465, remesh_intersect=60, batch_size=2)),
                   fine=Conf(start_epoch=12, point_render=Conf(radius=0.0041, remesh_intersect=120, batch_size=1))),
        sdf_net=Conf(multires=6),
        mlp_deformer=Conf(type="MLPTranslator", condlen=128, multires=6),
        render_net=Conf(type="RenderingNetwork_view_norm", multires_p=0, multires_x=0, multires_n=0, multires_v=4, condlen=256),
        loss_coarse=_loss(0.5, 2., 60., -10., 0.6),
        loss_medium=_loss(1.0, 3., 30., -1., 0.2),
        loss_fine=_loss(1.0, 4., 10., -1., 0.1, def_regu_w=0.07, sample_pix=6144))


def loose_config():
    """config_loose.conf as a delta of config.conf (the only differences, SURVEY D2: `diff config.conf config_loose.conf`): 600 epochs,
    milestones [30, 60, 240, 400], medium / fine from epochs 18 / 36, principal point and T not optimised, normal loss of the coarse
    stage switched off (normal_weight = -0.1)."""
    c = default_config()
    c['train']['nepoch'] = 600
    c['train']['opt_camera'] = Conf(focal_length=True, princeple_points=False, quat=False, T=False)
    c['train']['scheduler']['milestones'] = [30, 60, 240, 400]
    c['train']['medium']['start_epoch'] = 18
    c['train']['fine']['start_epoch'] = 36
    c['loss_coarse']['normal_weight'] = -0.1
    return c
