"""Checkpoint loader with the reference's name and return signature.

Mirrors /root/reference/pepper_variant/modules/python/models/ModelHander.py:18-44: reads
{'model_state_dict', 'hidden_size', 'gru_layers', 'epochs'}, strips 'module.' prefixes, returns
(model, hidden_size, gru_layers, epochs) -- the model being the HIP-backed TransducerGRU.
"""
import torch

from pepper_amd.variant.models.simple_model import TransducerGRU


class ModelHandler:
    @staticmethod
    def get_new_gru_model(image_features, gru_layers, hidden_size, num_classes, num_classes_type):
        return TransducerGRU(image_features, gru_layers, hidden_size, num_classes, num_classes_type,
                             bidirectional=True)

    @staticmethod
    def load_simple_model_for_training(model_path, image_features, num_classes, num_type_classes):
        checkpoint = torch.load(model_path, map_location='cpu')
        hidden_size = checkpoint['hidden_size']
        gru_layers = checkpoint['gru_layers']
        epochs = checkpoint['epochs']
        model = ModelHandler.get_new_gru_model(image_features=image_features, gru_layers=gru_layers,
                                               hidden_size=hidden_size, num_classes=num_classes,
                                               num_classes_type=num_type_classes)
        state = {}
        for k, v in checkpoint['model_state_dict'].items():
            state[k[7:] if k[0:7] == 'module.' else k] = v
        model.load_state_dict(state)
        return model, hidden_size, gru_layers, epochs
