"""FeatureExtraction stage operators (VGG / ResNet) on the HIP path.

Same class names, constructor signatures and state_dict keys as the reference's modules/feature_extraction.py
(VGG_FeatureExtractor :8-47, ResNet_FeatureExtractor :100-108, BasicBlock :165-199, ResNet :202-352); the layers
are parameter containers, the forward is a chain of implicit-GEMM conv launches with BatchNorm statistics fused
into the conv epilogue and BN-apply / residual / ReLU / max-pool fused into one elementwise pass.
RCNN_FeatureExtractor / GRCL / GRCL_unit (:50-162; selectable through opt.FeatureExtraction = "RCNN", used by no shipped
config) run on the same conv kernels plus the standalone BatchNorm / sigmoid / gating passes of mrn_amd.functional.
"""
import torch.nn as nn

from ._nn import conv_block, from_nhwc, require_no_grad, to_nhwc


class GRCL_unit(nn.Module):
    """one iteration of the gated recurrent conv layer (reference :146-161): five BatchNorms, a sigmoid gate, a gated sum"""

    def __init__(self, output_channel):
        super().__init__()
        self.BN_gfu = nn.BatchNorm2d(output_channel)
        self.BN_grx = nn.BatchNorm2d(output_channel)
        self.BN_fu = nn.BatchNorm2d(output_channel)
        self.BN_rx = nn.BatchNorm2d(output_channel)
        self.BN_Gx = nn.BatchNorm2d(output_channel)

    def forward(self, wgf_u, wgr_x, wf_u, wr_x):
        """all four NHWC: G = sigmoid(BN(wgf_u) + BN(wgr_x)); x = relu(BN(wf_u) + BN(BN(wr_x) * G))"""
        from ..functional import AddFn, AddReluFn, MulFn, SigmoidFn, batch_norm_nhwc
        G = SigmoidFn.apply(AddFn.apply(batch_norm_nhwc(wgf_u, self.BN_gfu), batch_norm_nhwc(wgr_x, self.BN_grx)))
        x_second = batch_norm_nhwc(MulFn.apply(batch_norm_nhwc(wr_x, self.BN_rx), G), self.BN_Gx)
        return AddReluFn.apply(batch_norm_nhwc(wf_u, self.BN_fu), x_second)


class GRCL(nn.Module):
    """Gated recurrent convolution layer (reference :112-143): the feed-forward terms wgf_u / wf_u of the input are computed once,
    the recurrent 1x1 / 3x3 convolutions run num_iteration times"""

    def __init__(self, input_channel, output_channel, num_iteration, kernel_size, pad):
        super().__init__()
        self.wgf_u = nn.Conv2d(input_channel, output_channel, 1, 1, 0, bias=False)
        self.wgr_x = nn.Conv2d(output_channel, output_channel, 1, 1, 0, bias=False)
        self.wf_u = nn.Conv2d(input_channel, output_channel, kernel_size, 1, pad, bias=False)
        self.wr_x = nn.Conv2d(output_channel, output_channel, kernel_size, 1, pad, bias=False)
        self.BN_x_init = nn.BatchNorm2d(output_channel)
        self.num_iteration = num_iteration
        self.GRCL = nn.Sequential(*[GRCL_unit(output_channel) for _ in range(num_iteration)])

    def forward_nhwc(self, u):
        from ..functional import batch_norm_nhwc
        wgf_u = conv_block(u, self.wgf_u, relu=False)
        wf_u = conv_block(u, self.wf_u, relu=False)
        x = batch_norm_nhwc(wf_u, self.BN_x_init, relu=True)
        for i in range(self.num_iteration):
            x = self.GRCL[i](wgf_u, conv_block(x, self.wgr_x, relu=False), wf_u, conv_block(x, self.wr_x, relu=False))
        return x

    def forward(self, input):
        return from_nhwc(self.forward_nhwc(to_nhwc(input)))


class RCNN_FeatureExtractor(nn.Module):
    """FeatureExtractor of GRCNN (reference :50-97): conv + 3 gated recurrent conv layers + 2x2 conv, [B,512,1,65] for 32x256 crops"""

    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        oc = [output_channel // 8, output_channel // 4, output_channel // 2, output_channel]
        self.output_channel = oc
        self.ConvNet = nn.Sequential(
            nn.Conv2d(input_channel, oc[0], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
            GRCL(oc[0], oc[0], num_iteration=5, kernel_size=3, pad=1), nn.MaxPool2d(2, 2),
            GRCL(oc[0], oc[1], num_iteration=5, kernel_size=3, pad=1), nn.MaxPool2d(2, (2, 1), (0, 1)),
            GRCL(oc[1], oc[2], num_iteration=5, kernel_size=3, pad=1), nn.MaxPool2d(2, (2, 1), (0, 1)),
            nn.Conv2d(oc[2], oc[3], 2, 1, 0, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True))

    def forward(self, input):
        from ..functional import MaxPoolFn, needs_grad
        from .. import ops
        c = self.ConvNet
        p22, p2_21 = ((2, 2), (2, 2), (0, 0)), ((2, 2), (2, 1), (0, 1))

        def pool(x, p):
            return MaxPoolFn.apply(x, *p) if (needs_grad(self, x)) else ops.maxpool_nhwc(x, *p)
        x = conv_block(to_nhwc(input), c[0], pool=p22)
        x = pool(c[3].forward_nhwc(x), p22)
        x = pool(c[5].forward_nhwc(x), p2_21)
        x = pool(c[7].forward_nhwc(x), p2_21)
        x = conv_block(x, c[9], c[10])
        return from_nhwc(x)


class VGG_FeatureExtractor(nn.Module):
    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        oc = [output_channel // 8, output_channel // 4, output_channel // 2, output_channel]
        self.output_channel = oc
        self.ConvNet = nn.Sequential(
            nn.Conv2d(input_channel, oc[0], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
            nn.Conv2d(oc[0], oc[1], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d(2, 2),
            nn.Conv2d(oc[1], oc[2], 3, 1, 1), nn.ReLU(True),
            nn.Conv2d(oc[2], oc[2], 3, 1, 1), nn.ReLU(True), nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(oc[2], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True),
            nn.Conv2d(oc[3], oc[3], 3, 1, 1, bias=False), nn.BatchNorm2d(oc[3]), nn.ReLU(True),
            nn.MaxPool2d((2, 1), (2, 1)),
            nn.Conv2d(oc[3], oc[3], 2, 1, 0), nn.ReLU(True),
        )

    def forward(self, input):
        c = self.ConvNet
        x = to_nhwc(input)
        p22, p21 = ((2, 2), (2, 2), (0, 0)), ((2, 1), (2, 1), (0, 0))
        x = conv_block(x, c[0], pool=p22)
        x = conv_block(x, c[3], pool=p22)
        x = conv_block(x, c[6])
        x = conv_block(x, c[8], pool=p21)
        x = conv_block(x, c[11], c[12])
        x = conv_block(x, c[14], c[15], pool=p21)
        x = conv_block(x, c[18])
        return from_nhwc(x)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward_nhwc(self, x):
        out = conv_block(x, self.conv1, self.bn1)
        res = x
        if self.downsample is not None:
            res = conv_block(x, self.downsample[0], self.downsample[1], relu=False)
        return conv_block(out, self.conv2, self.bn2, relu=True, residual=res)

    def forward(self, x):
        return from_nhwc(self.forward_nhwc(to_nhwc(x)))


class ResNet(nn.Module):
    def __init__(self, input_channel, output_channel, block, layers):
        super().__init__()
        ocb = [output_channel // 4, output_channel // 2, output_channel, output_channel]
        self.output_channel_block = ocb
        self.inplanes = output_channel // 8
        self.conv0_1 = nn.Conv2d(input_channel, output_channel // 16, 3, 1, 1, bias=False)
        self.bn0_1 = nn.BatchNorm2d(output_channel // 16)
        self.conv0_2 = nn.Conv2d(output_channel // 16, self.inplanes, 3, 1, 1, bias=False)
        self.bn0_2 = nn.BatchNorm2d(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool1 = nn.MaxPool2d(2, 2, 0)
        self.layer1 = self._make_layer(block, ocb[0], layers[0])
        self.conv1 = nn.Conv2d(ocb[0], ocb[0], 3, 1, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(ocb[0])
        self.maxpool2 = nn.MaxPool2d(2, 2, 0)
        self.layer2 = self._make_layer(block, ocb[1], layers[1])
        self.conv2 = nn.Conv2d(ocb[1], ocb[1], 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(ocb[1])
        self.maxpool3 = nn.MaxPool2d(2, (2, 1), (0, 1))
        self.layer3 = self._make_layer(block, ocb[2], layers[2])
        self.conv3 = nn.Conv2d(ocb[2], ocb[2], 3, 1, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(ocb[2])
        self.layer4 = self._make_layer(block, ocb[3], layers[3])
        self.conv4_1 = nn.Conv2d(ocb[3], ocb[3], 2, (2, 1), (0, 1), bias=False)
        self.bn4_1 = nn.BatchNorm2d(ocb[3])
        self.conv4_2 = nn.Conv2d(ocb[3], ocb[3], 2, 1, 0, bias=False)
        self.bn4_2 = nn.BatchNorm2d(ocb[3])

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * block.expansion))
        seq = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        seq += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def forward(self, x):
        x = to_nhwc(x)
        x = conv_block(x, self.conv0_1, self.bn0_1)
        x = conv_block(x, self.conv0_2, self.bn0_2, pool=((2, 2), (2, 2), (0, 0)))
        for blk in self.layer1:
            x = blk.forward_nhwc(x)
        x = conv_block(x, self.conv1, self.bn1, pool=((2, 2), (2, 2), (0, 0)))
        for blk in self.layer2:
            x = blk.forward_nhwc(x)
        x = conv_block(x, self.conv2, self.bn2, pool=((2, 2), (2, 1), (0, 1)))
        for blk in self.layer3:
            x = blk.forward_nhwc(x)
        x = conv_block(x, self.conv3, self.bn3)
        for blk in self.layer4:
            x = blk.forward_nhwc(x)
        x = conv_block(x, self.conv4_1, self.bn4_1)
        x = conv_block(x, self.conv4_2, self.bn4_2)
        return from_nhwc(x)


class ResNet_FeatureExtractor(nn.Module):
    def __init__(self, input_channel, output_channel=512):
        super().__init__()
        self.ConvNet = ResNet(input_channel, output_channel, BasicBlock, [1, 2, 5, 3])

    def forward(self, input):
        return self.ConvNet(input)
