"""create_model of the reference (models/__init__.py:3-29), inference subset."""
import logging


def create_model(opt):
    if opt.model == 'dec_vit':
        from .model_iid_dehazing import DECHLGVIT
        model = DECHLGVIT()
    else:
        raise NotImplementedError('model [%s] not implemented.' % opt.model)
    model.initialize(opt)
    logging.info("model [%s] was created" % (model.name()))
    return model
