"""Config object + click entry group for the LSTUR path.

Keeps the reference's entry points (/root/reference/settings.py): the ``main`` click
group with its five path options and the discarded ``--node-*`` options (:214-228),
``Config`` with derived path properties (:95-211), the ``pass_config`` decorator
(:231-238) and ``PythonLiteralOption`` (:12-21).  ``tf.gfile`` is replaced by ``os``.
Additions for the native path: ``device``, ``world_size``, ``seed`` keys.
"""
import ast
import os
import pprint

import click

os.environ.setdefault('LC_ALL', 'C.UTF-8')
os.environ.setdefault('LANG', 'C.UTF-8')

public_config = {}

_PATH_KEYS = ('input_training_data_path', 'input_validation_data_path', 'input_previous_model_path',
              'output_model_path', 'log_dir')
# option keys the `train` command of the LSTUR path defines (reference main.py:12-54)
_TRAIN_KEYS = ('task arch round days epochs batch_size training_step validation_step '
               'validation_impression testing_impression learning_rate learning_rate_decay gain id_keep '
               'window_size dropout negative_samples hidden_dim nonlocal_negative_samples '
               'enable_baseline title_filter_shape title_shape body_shape user_embedding_dim '
               'textual_embedding_dim textual_embedding_trainable debug background name '
               'pretrain_name enable_pretrain_encoder pretrain_encoder_trainable '
               'personal_embedding_dim news_encoder score_model body_sent_cnt body_sent_len '
               'body_filter_shape max_impression max_impression_pos max_impression_neg '
               'test_window_size vertical_embedding_dim subvertical_embedding_dim').split()
# keys of the other reference commands, accepted so a shared Config dict still loads
_OTHER_KEYS = ('evaluate_sessions target use_vertical pipeline_input lrd_on_epochs '
               'use_generator user_feature_size doc_feature_size use_vertical_type').split()
_NATIVE_KEYS = ('device', 'world_size', 'seed', 'ids', 'native_feed', 'native_load')


class PythonLiteralOption(click.Option):
    """Option parsed with ast.literal_eval after mapping '/' to '"' (shell-friendly quoting)."""

    def type_cast_value(self, ctx, value):
        if not isinstance(value, str):
            return value
        try:
            return ast.literal_eval(value.replace('/', '"'))
        except Exception:
            raise click.BadParameter(value)


class Config:
    __slots__ = list(_PATH_KEYS) + _TRAIN_KEYS + _OTHER_KEYS + list(_NATIVE_KEYS)

    def __init__(self, config, verbose=True):
        config.update(public_config)
        for key, value in config.items():
            if not key.startswith('node'):        # cluster-launcher residue is dropped
                setattr(self, key, value)
        if verbose:
            pprint.pprint(config)
            try:
                print('Input Training Data Path:')
                pprint.pprint(sorted(os.listdir(self.input_training_data_path)))
            except (FileNotFoundError, AttributeError):
                pass

    def _in(self, name):
        return os.path.join(self.input_training_data_path, name)

    training_data_input = property(lambda self: self._in('ClickData.tsv'))
    testing_data_input = property(lambda self: self._in('TestData.tsv'))
    title_embedding_input = property(lambda self: self._in('Vocab.tsv'))
    doc_meta_input = property(lambda self: self._in('DocMeta.tsv'))

    def _pair(self, root, stem, name):
        return (os.path.join(root, '{}{}.json'.format(stem, name)),
                os.path.join(root, '{}{}.pkl'.format(stem, name)))

    model_input = property(lambda self: self._pair(self.input_previous_model_path, 'model', self.pretrain_name))
    model_output = property(lambda self: self._pair(self.output_model_path, 'model', self.name))
    encoder_input = property(lambda self: self._pair(self.input_previous_model_path, 'encoder', self.pretrain_name))
    encoder_output = property(lambda self: self._pair(self.output_model_path, 'encoder', self.name))
    user_encoder_output = property(lambda self: self._pair(self.output_model_path, 'user.encoder', self.name))
    result_output = property(lambda self: os.path.join(self.output_model_path, 'model{}.tsv'.format(self.name)))
    log_output = property(lambda self: os.path.join(self.log_dir, 'log{}.txt'.format(self.name)))


@click.group(context_settings=dict(allow_extra_args=True, ignore_unknown_options=True))
@click.option('--input-training-data-path', default='../data')
@click.option('--input-validation-data-path', default='../data')
@click.option('--input-previous-model-path', default='../models')
@click.option('--output-model-path', default='../models')
@click.option('--log-dir', default='../logs')
@click.option('--node-count', default=1)
@click.option('--node-list-path', default='')
@click.option('--node-id', default=1)
@click.option('--node-name', default='')
def main(**_config):
    global public_config
    public_config = _config
    os.makedirs(_config['output_model_path'], exist_ok=True)
    os.makedirs(_config['log_dir'], exist_ok=True)
    return 0


def pass_config(f):
    def __wrapper__(**_config):
        return f(Config(_config))

    __wrapper__.__name__ = f.__name__
    return __wrapper__
