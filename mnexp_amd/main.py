"""`train` command of the LSTUR path on the native MI355X model.

Keeps the reference's entry point and option names (/root/reference/main.py:11-96, 442):
    python -m mnexp_amd.main --input-training-data-path DATA --output-model-path OUT --log-dir LOGS \\
        train -t Seq2VecPaperSoftmaxId -a igru -e 2 -b 64 -s 100 -w 50 --title-shape 30 ...
Every option is also settable as MSN_NEWS_<OPTION> (auto_envvar_prefix).  Added for the native path:
--seed, --ids (feed doc-id batches and keep titles resident on the GPU instead of expanding them on the
host), --native-feed (C++ sample feed, bit-identical batches).  Under `torch.distributed.run` each rank trains on its own shard of users (rank r takes users
r, r+N, ...) with RCCL gradient averaging (mnexp_amd/dist.py).
"""
import logging

import click

from . import settings, task, utils

_TRAIN_OPTIONS = [
    (('-t', '--task'), dict(default='Seq2VecPaperSoftmaxId')),
    (('-a', '--arch'), dict(default='igru')),
    (('-r', '--round'), dict(default=6)),
    (('-y', '--days'), dict(default=30)),
    (('-e', '--epochs'), dict(default=10)),
    (('-b', '--batch-size'), dict(default=100)),
    (('-s', '--training-step'), dict(default=10000)),
    (('-v', '--validation-step'), dict(default=1000)),
    (('-cvi', '--validation-impression'), dict(default=1000)),
    (('-i', '--testing-impression'), dict(default=1000)),
    (('-l', '--learning-rate'), dict(default=0.001)),
    (('-cld', '--learning-rate-decay'), dict(default=0.2)),
    (('-g', '--gain'), dict(default=1.0)),
    # the LSTUR paper's user-embedding masking probability (reference: the `cook` command's --id-keep, main.py:134,
    # task/cook.py:139-142), offered to the Paper tasks: a training sample keeps its user embedding with this probability
    (('--id-keep',), dict(default=1.0)),
    (('-w', '--window-size'), dict(default=10)),
    (('-d', '--dropout'), dict(default=0.2)),
    (('-n', '--negative-samples'), dict(default=4)),
    (('-h', '--hidden-dim'), dict(default=400)),
    (('-nn', '--nonlocal-negative-samples'), dict(default=0)),
    (('-ceb', '--enable-baseline'), dict(is_flag=True)),
    (('--title-filter-shape',), dict(default=(400, 3), nargs=2)),
    (('--title-shape',), dict(default=20)),
    (('--body-shape',), dict(default=200)),
    (('--user-embedding-dim',), dict(default=200)),
    (('--textual-embedding-dim',), dict(default=300)),
    (('--textual-embedding-trainable',), dict(is_flag=True)),
    (('--debug',), dict(is_flag=True)),
    (('--background',), dict(is_flag=True)),
    (('--name',), dict(default='')),
    (('-p', '--pretrain-name'), dict(default='')),
    (('-cep', '--enable-pretrain-encoder'), dict(is_flag=True)),
    (('-cpt', '--pretrain-encoder-trainable'), dict(is_flag=True)),
    (('--personal-embedding-dim',), dict(default=20)),
    (('--news-encoder',), dict(default='cnnatt')),
    (('--score-model',), dict(default='dot')),
    (('--body-sent-cnt',), dict(default=50)),
    (('--body-sent-len',), dict(default=30)),
    (('--body-filter-shape',), dict(default=(400, 3), nargs=2)),
    (('--max-impression',), dict(default=200)),
    (('--max-impression-pos',), dict(default=7)),
    (('--max-impression-neg',), dict(default=200)),
    (('--test-window-size',), dict(default=100)),
    (('--vertical-embedding-dim',), dict(default=10)),
    (('--subvertical-embedding-dim',), dict(default=20)),
    # native-path additions
    (('--seed',), dict(default=0)),
    (('--ids',), dict(is_flag=True, help='feed doc-id batches; titles stay resident on the GPU')),
    (('--native-feed',), dict(is_flag=True, help='C++ sample feed (liblstur_data.so): the same batches as the Python '
                                                  'generators for np.random.seed(--seed), implies --ids')),
    (('--native-load/--python-load',), dict(default=True, help='parse DocMeta / ClickData / Vocab.tsv with the C++ loaders '
                                                                '(liblstur_data.so; same tables, same errors) or in Python')),
]


def _with_options(f):
    for names, kw in reversed(_TRAIN_OPTIONS):
        f = click.option(*names, **kw)(f)
    return f


def run_training(config, task_handler=None):
    """Epoch loop with the reference's shape (main.py:66-94): build_model, fit_generator, callback,
    evaluate_generator (failures swallowed like the reference), callback_valid, save_model."""
    if task_handler is None:
        task_handler = task.get(config)
    native_feed = bool(getattr(config, 'native_feed', False))
    use_ids = bool(getattr(config, 'ids', False)) or native_feed
    feed = task_handler.native_feed() if (native_feed and hasattr(task_handler, 'native_feed')) else None
    if feed is not None:
        training_data = feed
    else:
        training_data = task_handler.train_ids if use_ids else task_handler.train
    model = None
    for epoch in range(config.epochs):
        logging.info('[+] start epoch {}'.format(epoch))
        model = task_handler.build_model(epoch)
        history = model.fit_generator(training_data, task_handler.training_step, epochs=epoch + 1,
                                      initial_epoch=epoch, verbose=1 if config.debug and not config.background else 2,
                                      ids_form=use_ids)
        utils.logging_history(history)
        if hasattr(task_handler, 'callback'):
            task_handler.callback(epoch)
        try:
            evaluations = model.evaluate_generator(task_handler.valid, task_handler.validation_step,
                                                   verbose=1 if config.debug and not config.background else 2)
            utils.logging_evaluation(dict(zip(model.metrics_names, evaluations)))
        except Exception:        # the reference swallows evaluation failures (main.py:88-89)
            pass
        if hasattr(task_handler, 'callback_valid'):
            task_handler.callback_valid(epoch)
        logging.info('[-] finish epoch {}'.format(epoch))
    task_handler.save_model()
    return task_handler


@settings.main.command(context_settings=dict(allow_extra_args=True, ignore_unknown_options=True))
@_with_options
@settings.pass_config
def train(config):
    from . import dist
    dist.maybe_init_from_env()
    # under data parallelism every rank trains on its shard of users; the callback's test impressions are dealt out over the
    # ranks and their metric sums all-reduced (the single-GPU numbers); only rank 0 writes the log file and the console
    handlers = [logging.FileHandler(config.log_output), logging.StreamHandler()] if dist.rank() == 0 else [logging.NullHandler()]
    logging.basicConfig(format='%(asctime)s : %(levelname)s : %(message)s', level=logging.INFO, handlers=handlers)
    run_training(config)
    return 0


if __name__ == '__main__':
    exit(settings.main(auto_envvar_prefix='MSN_NEWS'))
