classdef KsysidHip < Ksysid
    %KsysidHip: the reference's Ksysid with the EDMD fit on an MI355X (libkoopman_hip.so through kp_mex).
    %   Drop-in: replace `Ksysid(` by `KsysidHip(` in example_sysid.m / Ksysid_setup.m; every property, Name/Value
    %   argument and method signature is the parent's.  Overridden: get_Koopman (Ksysid.m:987-1092: per-row lift loop,
    %   Px'Px / Px'Py, mldivide / lasso QP) and train_models for a vector of lasso values (:1372-1387: the snapshots
    %   are lifted once and all values solved as one batch).  Everything else (symbolic dictionary for display, scaling,
    %   snapshot pairs, pca, get_model/get_BLmodel/get_NLmodel, val_*) runs as in the parent on the K returned here.
    %   Not executed in the build image (no MATLAB there); the Python class koopman_realizations_amd.Ksysid is the
    %   tested mirror of exactly these calls.

    properties
        hip;    % struct: ctx, basis (uint64 handles of kp_mex), W
    end

    methods
        function obj = KsysidHip( data4sysid , varargin )
            obj = obj@Ksysid( data4sysid , varargin{:} );
            obj.hip.ctx = kp_mex( 'create' , 0 );
            obj.hip.basis = kp_mex( 'basis_create' , obj.hip.ctx , obj.hip_descriptor );
            d = kp_mex( 'basis_dims' , obj.hip.basis );
            obj.hip.W = d(4);
            assert( d(3) == obj.params.N , 'device dictionary and params.N disagree' );
            % lift.econ_full / econ_full_input (Ksysid.m:1594-1618) as device calls
            obj.lift.econ_full = @(zeta) kp_mex( 'lift' , obj.hip.ctx , obj.hip.basis , 1 , zeta' , [] )';
            if ~strcmp( obj.model_type , 'linear' )
                obj.lift.econ_full_input = @(zeta,u) kp_mex( 'lift' , obj.hip.ctx , obj.hip.basis , 2 , zeta' , u' )';
            end
        end

        function d = hip_descriptor( obj )
            % the dictionary as data (def_observables, Ksysid.m:455-536): blocks in obs_type order
            nv = obj.params.nzeta + obj.params.m * strcmp( obj.model_type , 'nonlinear' );
            bt = int32([]); bc = int32([]); ex = uint8([]); cen = [];
            for i = 1 : length( obj.obs_type )
                deg = obj.obs_degree(i);
                switch obj.obs_type{i}
                    case 'poly'       % monomial order of def_polyLift (:645-648): partitions per total degree
                        e = [];
                        for dgr = 1 : deg
                            e = [ e ; partitions( dgr , ones(1,nv) ) ]; %#ok<AGROW>
                        end
                        e = e( nv+1 : end , : );      % the first nv monomials are zeta itself (:488)
                        bt(end+1) = 0; bc(end+1) = size(e,1); ex = [ ex , uint8(e') ]; %#ok<AGROW>
                    case 'fourier'
                        bt(end+1) = 1; bc(end+1) = deg; %#ok<AGROW>
                    case 'gaussian'
                        bt(end+1) = 2; bc(end+1) = deg; cen = [ cen , obj.params.gaussian_centres ]; %#ok<AGROW>
                    otherwise
                        error( 'KsysidHip: obs_type %s - see INTEGRATION.md for the hermite / fourier_sparser blocks' , obj.obs_type{i} );
                end
            end
            mt = find( strcmp( obj.model_type , { 'linear' , 'bilinear' , 'nonlinear' } ) ) - 1;
            pcs = [];
            if obj.dim_red
                pcs = obj.basis.pcs;     % Nfull x k (get_econ_observables, Ksysid.m:1498-1516)
            end
            d = struct( 'model_type' , int32(mt) , 'nzeta' , int32(obj.params.nzeta) , 'm' , int32(obj.params.m) , ...
                        'block_type' , bt , 'block_count' , bc , 'poly_exps' , ex , 'gauss_centres' , cen , 'pcs' , pcs );
        end

        function koopData = get_Koopman( obj , snapshotPairs , varargin )
            % Ksysid.m:987-1092
            if length( varargin ) == 1
                lasso = varargin{1};        % t = lasso * N is formed inside the library (:996)
            else
                lasso = 1e4;                % :999
            end
            if obj.lasso >= 1e6             % :1068 tests the PROPERTY
                lasso = Inf;
            end
            % the context's resident snapshot object, refilled in place (no device allocation per call; staged transfer)
            s = kp_mex( 'snapshots_resident' , obj.hip.ctx , snapshotPairs.alpha , snapshotPairs.beta , snapshotPairs.u );
            K = kp_mex( 'fit' , obj.hip.ctx , obj.hip.basis , s , lasso );
            N = obj.params.N;
            Px = kp_mex( 'lift' , obj.hip.ctx , obj.hip.basis , 2 , snapshotPairs.alpha , snapshotPairs.u );
            Py = kp_mex( 'lift' , obj.hip.ctx , obj.hip.basis , 2 , snapshotPairs.beta , snapshotPairs.u );
            koopData.K = K;                          % :1084-1091
            koopData.Px = Px( : , 1 : N );
            koopData.Py = Py( : , 1 : N );
            koopData.u = snapshotPairs.u;
            koopData.alpha = snapshotPairs.alpha;
        end

        function obj = train_models( obj , lasso )
            % Ksysid.m:1344-1389; a vector of lasso values is ONE device call (lift once, values batched)
            if nargin < 2
                lasso = obj.lasso;
            end
            if length( lasso ) == 1
                obj = train_models@Ksysid( obj , lasso );
                return;
            end
            sp = obj.snapshotPairs;
            s = kp_mex( 'snapshots_resident' , obj.hip.ctx , sp.alpha , sp.beta , sp.u );
            Ks = kp_mex( 'fit' , obj.hip.ctx , obj.hip.basis , s , lasso(:)' );
            N = obj.params.N;
            Px = kp_mex( 'lift' , obj.hip.ctx , obj.hip.basis , 2 , sp.alpha , sp.u );
            Py = kp_mex( 'lift' , obj.hip.ctx , obj.hip.basis , 2 , sp.beta , sp.u );
            obj.candidates = cell( length(lasso) , 1 );
            for i = 1 : length( lasso )                  % :1372-1387
                kd = struct( 'K' , Ks(:,:,i) , 'Px' , Px(:,1:N) , 'Py' , Py(:,1:N) , 'u' , sp.u , 'alpha' , sp.alpha );
                if strcmp( obj.model_type , 'linear' )
                    obj.candidates{i} = obj.get_model( kd );
                elseif strcmp( obj.model_type , 'bilinear' )
                    obj.candidates{i} = obj.get_BLmodel( kd );
                else
                    obj.candidates{i} = obj.get_NLmodel( kd );
                end
                obj.candidates{i}.lasso = lasso(i);
            end
            obj.koopData = kd;
        end

        function delete_hip( obj )
            kp_mex( 'basis_destroy' , obj.hip.basis );
            kp_mex( 'destroy' , obj.hip.ctx );
        end
    end
end
