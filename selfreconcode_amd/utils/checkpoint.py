"""Checkpoint format of the reference (utils/utils.py:257-316): one dict with
`epoch`, `model_state_dict` (keys sdf.linK.weight_g/_v/bias, deformer.defs.0.linK.*, netRender.linK.*,
deformer.defs.1.{ws,b_min,b_max,Js,init_pose}, engine.*), the four camera tensors, `poses`, `trans`, `shape`,
`dcond`, `rcond`.  Files written by the reference load here and vice versa."""
import torch


def save_model(name, epoch, optNet, dataset):
    outdic = {"epoch": epoch, "model_state_dict": optNet.state_dict()}
    outdic.update(dataset.camera_params)
    outdic.update({'poses': dataset.poses, 'trans': dataset.trans, 'shape': dataset.shape, 'dcond': dataset.conds[0], 'rcond': dataset.conds[1]})
    torch.save(outdic, name)


def load_model(name, optNet, dataset, device, subsdfmodel=None, model_rm_prefix=None):
    saved = torch.load(name, map_location='cpu')
    state = {k: v for k, v in saved["model_state_dict"].items() if 'engine.' not in k}            # engine buffers are rebuilt (:268)
    if model_rm_prefix:
        state = {k: v for k, v in state.items() if not any(k.startswith(p) for p in model_rm_prefix)}
    if subsdfmodel is not None:
        sdf_model = torch.load(subsdfmodel, map_location='cpu')
        state = {k: v for k, v in state.items() if 'sdf.' not in k}
        state.update({'sdf.' + k: v for k, v in sdf_model.items()})
    state = {k: v for k, v in state.items() if 'deformer.defs.1.ws' not in k}                        # the skinning volume is never restored (:286)
    optNet.load_state_dict(state, strict=False)
    optNet = optNet.to(device)
    dev = dataset.poses.device
    if 'dcond' in saved:
        dataset.conds[0] = saved['dcond'].to(dev).requires_grad_()
    if 'rcond' in saved:
        dataset.conds[1] = saved['rcond'].to(dev).requires_grad_()
    for attr in ('poses', 'trans', 'shape'):
        grad = getattr(dataset, attr).requires_grad
        setattr(dataset, attr, saved[attr].detach().to(dev).requires_grad_(grad))
    assert dataset.frame_num <= dataset.poses.shape[0] and dataset.frame_num <= dataset.trans.shape[0]
    dataset.camera_params = {k: saved[k].detach().to(dev).requires_grad_(v.requires_grad) for k, v in dataset.camera_params.items()}
    return optNet, dataset


def set_hierarchical_config(conf, name, optNet, dataloader, resolutions):
    """Stage switch coarse -> medium -> fine (utils/utils.py:237-255), same signature and return value: a DataLoader over the same
    dataset / sampler with the stage's batch size (None stays None: the synthetic sequence has no loader), the stage's loss /
    point-render configuration left PENDING on the network (`next_conf`, `next_train_conf`: OptimNetwork adopts them at its next
    scheduled remesh, network.py:464) and a new Seg3dLossless engine at the stage's resolution pyramid, in place at once."""
    from ..MCAcc import Seg3dLossless
    batch_size = conf.get_int('train.' + name + '.point_render.batch_size')
    if dataloader is not None:
        dataloader = torch.utils.data.DataLoader(dataloader.dataset, batch_size, sampler=dataloader.sampler, num_workers=dataloader.num_workers)
    optNet.next_conf = conf.get_config('loss_' + name)
    optNet.next_train_conf = conf.get_config('train.' + name)
    optNet.engine = Seg3dLossless(query_func=None, b_min=optNet.engine.b_min, b_max=optNet.engine.b_max, resolutions=resolutions,
                                  align_corners=False, balance_value=0.0,
                                  use_cuda_impl=getattr(optNet.engine, 'use_cuda_impl', True)).to(optNet.engine.b_min.device)
    return optNet, dataloader
