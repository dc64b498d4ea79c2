# mmcv-style config (same sections and keys as the reference's config/*.py, which load unchanged through
# `python -m mrn_amd.tiny_train --config <file>`): a short synthetic MRN run of the CRNN family for smoke-testing the driver:
#     python -m mrn_amd.tiny_train --config config/crnn_mrn_synthetic.py --synthetic
common = dict(exp_name="CRNN_MRN_synthetic", il="mrn", memory="random", memory_num=2000, batch_max_length=25, imgH=32, imgW=256,
              manual_seed=111, start_task=0)
model = dict(model_name="CRNN", Transformation="None", FeatureExtraction="VGG", SequenceModeling="BiLSTM", Prediction="CTC",
             num_fiducial=20, input_channel=4, output_channel=512, hidden_size=256)
optimizer = dict(schedule="super", optimizer="adam", lr=0.0005, sgd_momentum=0.9, sgd_weight_decay=0.000001, milestones=[2000, 4000],
                 lrate_decay=0.1, rho=0.95, eps=1e-8, lr_drop_rate=0.1)
train = dict(saved_model="", Aug="None", workers=0, lan_list=["Chinese", "Latin", "Japanese"], valid_datas=["synthetic"],
             select_data=["synthetic"], NED=True, batch_size=64, num_iter=20, val_interval=10, grad_clip=5)
