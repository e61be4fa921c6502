function dump_reference_vectors(in_file, out_file)
% DUMP_REFERENCE_VECTORS  Run the REFERENCE's decoder core on the committed fixture LLRs and save what it returns.
%
% The one route by which the build's decoder oracle can be pinned to the reference's own arithmetic: the reference decodes with
% MathWorks' closed-source comm.LDPCDecoder (NRLDPCDecoder.m:120, :265), the build image holds no MATLAB, and the repository
% holds no decoder vectors (SURVEY.md section 8c) -- so every decoder check in this repository compares with a restatement of
% the documented algorithm (oracle/nrldpc_oracle.c: orc_decode_bp_flood), "parity unpinned".  On a machine with MATLAB, the
% Communications Toolbox and a checkout of robmaunder/ldpc-3gpp-matlab on the path:
%
%     >> addpath('<checkout of ldpc-3gpp-matlab>'); cd <this repo>/matlab; dump_reference_vectors
%
% reads  ../tests/golden/reference_inputs.mat   (tests/golden/export_reference_inputs.py: the LLRs of nmsq_golden.npz)
% writes ../tests/golden/reference_outputs.mat  (hard decisions and NumIterations per case, column and iteration cap)
%
% and `python -m pytest tests/test_reference_dump.py` then compares the oracle with that file bit for bit (the test is skipped
% while the file is absent).  Commit the file: row (c) of the coverage table leaves "unpinned" with it.
%
% What runs is exactly the reference's own construction: H = get_pcm(get_3gpp_base_graph(BG, i_LS), Z) as NRLDPC.get.H does
% (NRLDPC.m:433-440), comm.LDPCDecoder('ParityCheckMatrix',H,'MaximumIterationCount',iterations,
% 'IterationTerminationCondition','Parity check satisfied') as NRLDPCDecoder.m:120 does -- plus 'NumIterationsOutputPort' so
% that the number of sweeps is saved too -- and step(hDec, cw_tilde) per column as NRLDPCDecoder.m:265 does.
if nargin < 1, in_file = fullfile('..', 'tests', 'golden', 'reference_inputs.mat'); end
if nargin < 2, out_file = fullfile('..', 'tests', 'golden', 'reference_outputs.mat'); end
S = load(in_file);
cases = S.cases;
results = struct('name', {}, 'BG', {}, 'Z', {}, 'iterations', {}, 'hard', {}, 'num_iterations', {});
for k = 1:numel(cases)
    c = cases(k);
    BG = double(c.BG); Z = double(c.Z); its = double(c.iterations(:).');
    i_LS = get_3gpp_set_index(Z);                          % NRLDPC.m:428-430
    H = get_pcm(get_3gpp_base_graph(BG, i_LS), Z);         % NRLDPC.m:433-440
    if BG == 1, K = 22 * Z; else, K = 10 * Z; end          % NRLDPC.m:414-425
    llr = double(c.llr);                                    % (ncols*Z) x batch, one cw_tilde per column
    batch = size(llr, 2);
    hard = zeros(K, batch, numel(its));
    num_iterations = zeros(batch, numel(its));
    for t = 1:numel(its)
        hDec = comm.LDPCDecoder('ParityCheckMatrix', H, 'MaximumIterationCount', its(t), ...
            'IterationTerminationCondition', 'Parity check satisfied', 'NumIterationsOutputPort', true);
        for b = 1:batch
            [bits, n] = step(hDec, llr(:, b));              % NRLDPCDecoder.m:265
            hard(:, b, t) = double(bits);
            num_iterations(b, t) = double(n);
        end
        release(hDec);
    end
    results(k).name = c.name; results(k).BG = BG; results(k).Z = Z; results(k).iterations = its;
    results(k).hard = uint8(hard); results(k).num_iterations = int32(num_iterations);
    fprintf('%s: BG%d Z=%d, %d columns, caps %s, mean sweeps %s\n', c.name, BG, Z, batch, mat2str(its), mat2str(mean(num_iterations, 1), 4));
end
v = ver('comm'); r = version;
toolbox_version = ''; if ~isempty(v), toolbox_version = v(1).Version; end
matlab_release = r;
save(out_file, 'results', 'toolbox_version', 'matlab_release', '-v7');
fprintf('wrote %s\n', out_file);
end
