"""Polish checkpoint loader.  Mirrors /root/reference/pepper/modules/python/models/ModelHander.py:21-26,88-113
(get_new_gru_model, load_simple_model_for_training -> (model, hidden_size, gru_layers, epochs))."""
import torch

from pepper_amd.polish.models.simple_model import TransducerGRU


class ModelHandler:
    @staticmethod
    def get_new_gru_model(input_channels, image_features, gru_layers, hidden_size, num_classes=5):
        return TransducerGRU(input_channels, image_features, gru_layers, hidden_size, num_classes, bidirectional=True)

    @staticmethod
    def load_simple_model_for_training(model_path, input_channels, image_features, seq_len, num_classes):
        checkpoint = torch.load(model_path, map_location='cpu')
        hidden_size = checkpoint['hidden_size']
        gru_layers = checkpoint['gru_layers']
        epochs = checkpoint['epochs']
        model = ModelHandler.get_new_gru_model(input_channels=input_channels, image_features=image_features,
                                               gru_layers=gru_layers, hidden_size=hidden_size,
                                               num_classes=num_classes)
        state = {}
        for k, v in checkpoint['model_state_dict'].items():
            state[k[7:] if k[0:7] == 'module.' else k] = v
        model.load_state_dict(state)
        return model, hidden_size, gru_layers, epochs
